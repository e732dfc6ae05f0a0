set -u
OUT=gpurun_out/r02_b; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log
timeout 600 python tools/gemm_bench.py 8 variants > $OUT/gemm_bench_variants.txt 2>&1; tail -3 $OUT/gemm_bench_variants.txt
