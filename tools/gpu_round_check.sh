cd tools
for ps in 0 512 1024; do
echo "PERSIST=$ps"
echo "  P=16 K=128 HBM: $(PASTIX_AMD_PERSIST=$ps ./bench_update 8192 16 128 4096 | tail -1 | sed 's/.*launch, //')"
echo "  P=8  K=96  HBM: $(PASTIX_AMD_PERSIST=$ps ./bench_update 8192 8 96 4096 | tail -1 | sed 's/.*launch, //')"
echo "  P=4  K=96  HBM: $(PASTIX_AMD_PERSIST=$ps ROWS=1024 ./bench_update 8192 4 96 16384 | tail -1 | sed 's/.*launch, //')"
echo "  P=16 K=128 cache: $(PASTIX_AMD_PERSIST=$ps ./bench_update 8192 16 128 64 | tail -1 | sed 's/.*launch, //')"
done
