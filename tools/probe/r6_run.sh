mkdir -p gpurun_out/r6
timeout 1500 python tools/knob_sweep.py C2 C3 C5 2>&1 | grep -v amdgpu.ids > gpurun_out/r6/knob_sweep_maxilp.txt; cat gpurun_out/r6/knob_sweep_maxilp.txt | cut -c1-100
