set -x
timeout 300 python tools/experiments/ab_bitwise.py wide_direct_w=0 wide_direct_w=1 2>&1 | tail -13
timeout 600 python tools/conv_bench.py --images 8000 --iters 10 --rounds 5 --only D3,D4,P4 --sparse-input --nores --ab "wide_direct_w=0,wide_direct_w=1" 2>&1 | tail -8
BMI_WIDE_PERSIST=0 timeout 600 python tools/conv_bench.py --images 8000 --iters 10 --rounds 5 --only D3,D4,P4 --sparse-input --nores --ab "wide_direct_w=0,wide_direct_w=1" 2>&1 | tail -8
