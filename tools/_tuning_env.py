"""Developer tools only: PCGRL_* environment variables -> the tuning overrides of the Python binding (`_lib.TUNING_OVERRIDES`).

The library (libpcgrl_hip.so) reads no environment variables: its developer switches are the `pcgrl_tuning` struct of
include/pcgrl_hip.h, set per handle with pcgrl_set_tuning.  The measurement scripts in tools/ (sweeps, A/B runs under bench.py)
are driven from shell loops, so they keep the old variable names and translate them here:

    import _tuning_env; _tuning_env.apply()          # before the first BatchedPcgrlEnv is made
"""
import os

ENV_TO_FIELD = {
    "PCGRL_NO_FUSED": "no_fused", "PCGRL_FUSED_ZELDA": "fused_zelda", "PCGRL_STEP_EPB": "step_epb", "PCGRL_NO_INC": "no_inc",
    "PCGRL_INLINE_RESET": "inline_reset", "PCGRL_PAIR_MIN": "pair_min", "PCGRL_NO_WIDE": "no_wide", "PCGRL_WIDE_WAVES": "wide_waves",
    "PCGRL_WIDE_GRID": "wide_grid", "PCGRL_WIDE_PAIRS": "wide_pairs", "PCGRL_WIDE_FEW": "wide_few", "PCGRL_SOK_GENERIC": "sok_generic",
    "PCGRL_SOK_HARD_CAP": "sok_hard_cap", "PCGRL_SOK_SPAWN": "sok_spawn", "PCGRL_MD_ONLY_AGENT": "md_only_agent",
    "PCGRL_SMB_LDS_HEAP": "smb_lds_heap", "PCGRL_STEP_FPW": "full_per_wave", "PCGRL_STEP_IPW": "inc_per_wave", "PCGRL_WIDE_SPIN": "wide_spin", "PCGRL_STEP_PRIO": "step_prio", "PCGRL_NO_TOUCH": "no_touch", "PCGRL_TOUCH_TIGHT": "touch_tight", "PCGRL_STEP_PAIR": "step_pair",
}


def apply(environ=None):
    from gym_pcgrl_amd import _lib
    environ = os.environ if environ is None else environ
    for var, field in ENV_TO_FIELD.items():
        if var in environ:
            _lib.TUNING_OVERRIDES[field] = int(environ[var])
    return dict(_lib.TUNING_OVERRIDES)
