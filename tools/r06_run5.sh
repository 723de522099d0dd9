mkdir -p gpurun_out/r06
timeout 900 python tools/debug_slab.py > gpurun_out/r06/debug_slab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/debug_slab.txt | head -300
