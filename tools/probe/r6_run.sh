python tools/probe/collector_floor.py 2>&1 | grep -v amdgpu.ids | tail -3
