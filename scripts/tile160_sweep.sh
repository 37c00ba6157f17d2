SH="3150,3072,768,0,0,g;3150,3072,768,0,1,a;3150,2304,768,0,0,b;6272,1536,384,0,0,g;6272,1536,384,0,1,a;4192,3072,768,0,0,g;4192,3072,768,0,1,a;4192,2304,768,0,0,b;1568,3072,768,0,0,g;3150,768,3072,0,0,br;6272,384,1536,0,0,br"
for i in 1 2; do
for t in "64,64" "160,128" ""; do echo "== MVLT_TILE=$t"; if [ -n "$t" ]; then MVLT_TILE=$t SHAPES="$SH" python scripts/bench_gemm_shape.py 2>/dev/null; else SHAPES="$SH" python scripts/bench_gemm_shape.py 2>/dev/null; fi; done
done
