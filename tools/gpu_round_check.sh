cd tools
for cfg in "128 1 128" "24 1 300" "24 8192 300" "40 8192 300" "64 8192 300" "128 2048 1000"; do ./bench_diag $cfg | tail -1; done
