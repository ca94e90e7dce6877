for a in 0 2 3; do for t in 0 1 9 10; do echo "act $a tile $t"; BENCH_ACT=$a SR_IGEMM_TILE=$t python tools/bench_igemm.py 65536 1 1 320 2560 1 2>/dev/null; done; done
for a in 0 2; do echo "act $a 32^2 level"; BENCH_ACT=$a python tools/bench_igemm.py 16384 1 1 640 5120 1 2>/dev/null;  BENCH_ACT=$a python tools/bench_igemm.py 4096 1 1 1280 10240 1 2>/dev/null; done
