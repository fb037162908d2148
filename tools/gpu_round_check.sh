( ./tools/bench_update 8192 16 128 64
NOFAST=1 ./tools/bench_update 8192 16 128 64
STRUCT=1 ROWS=40960 ./tools/bench_update 16384 16 128 16
NOFAST=1 STRUCT=1 ROWS=40960 ./tools/bench_update 16384 16 128 16 ) 2>&1
