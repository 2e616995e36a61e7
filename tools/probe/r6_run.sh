mkdir -p gpurun_out/r6
(timeout 600 python tools/probe/collector_graph.py; timeout 600 python tools/probe/collector_graph.py zelda-wide-v0 wide) 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/r6/collector_graph.txt; cat gpurun_out/r6/collector_graph.txt
