cd tools
echo "cache-fed (pool 64):"; ./bench_update 8192 16 128 64 | tail -1; NWV=8 ./bench_update256 4096 16 128 64 | tail -2; NWV=16 ./bench_update256 4096 16 128 64 | tail -2
echo "HBM-fed (pool 4096):"; ./bench_update 8192 16 128 4096 | tail -1; NWV=8 ./bench_update256 4096 16 128 4096 | tail -1; NWV=16 ./bench_update256 4096 16 128 4096 | tail -1
