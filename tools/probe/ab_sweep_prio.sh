# same-box A/B of prebuilt libraries (tools/exp_build_local.py): bash tools/probe/ab_sweep_prio.sh "<so> <so> ..." "<workloads>"
mkdir -p gpurun_out/r5g
SOS=${1:-"gym_pcgrl_amd/lib/libexp_noprio.so gym_pcgrl_amd/lib/libpcgrl_hip.so"}
WLS=${2:-"C5 C5b"}
for rep in 1 2; do for so in $SOS; do echo "== $so"; PCGRL_HIP_SO=$so python tools/knob_sweep.py --defaults $WLS 2>&1 | grep -v amdgpu.ids; done; done
