set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "trunk" 2>&1 | tail -5
for e in 0 1; do echo "A2F_TRUNK_EXACT=$e"; A2F_TRUNK_EXACT=$e python tools/bench_trunk.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/trunk_ab.txt
cat gpurun_out/trunk_ab.txt
