for t in 128,128 128,96 64,128 64,96 64,64; do
  echo "== tile $t"
  MVLT_TILE=$t timeout 200 python scripts/bench_gemm.py 2>&1 | grep -E "TF/s" | awk -v t=$t '{print t, $0}' | cut -c1-100
done
