export TMPDIR=/tmp
for d in 0 1 2 3 4; do echo "DBG=$d"; ISC_GEMM_DEBUG=$d timeout -k 10 100 python scripts/quick_gemm_bench.py 2>&1 | grep -v amdgpu; done
