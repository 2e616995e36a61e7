mkdir -p gpurun_out/r6
(python tools/probe/sub_diag.py C2 C3 C2 C3 C3 C2 C3; PCGRL_STEP_EPB=256 python tools/probe/sub_diag.py C2 C3 C2 C3; python tools/probe/sub_batches_seq.py C2:2 C3:2 C2:2 C3:2 C2:3 C3:3) 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/sub_diag2.txt; cat gpurun_out/r6/sub_diag2.txt
