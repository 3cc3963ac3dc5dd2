mkdir -p gpurun_out/r4z
: > gpurun_out/r4z/stamps.txt
for v in study abl1 abl2 abl3; do
  echo "== VK_LIB=$v" >> gpurun_out/r4z/stamps.txt
  VK_LIB=$v STAMP_ONLY=GELU timeout -k 10 200 python3 tools/stamp_gemm.py >> gpurun_out/r4z/stamps.txt 2>&1 || exit 1
done
cat gpurun_out/r4z/stamps.txt
