python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm or prefetch" 2>&1 | tail -2
run() { python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config']['loss'])"; }
echo on; run; echo off; MVLT_WEIGHT_PREFETCH=0 run; echo on; run; echo off; MVLT_WEIGHT_PREFETCH=0 run
