# A/B of launch-shape tunables on ONE box (boxes differ by 3-4 %): every line is `graph: 20 steps in X s`
run() { echo -n "$1: "; env $1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-modes --no-cpu-baseline 2>&1 >/dev/null | grep -o "graph: 20 steps in [0-9.]* s"; }
run "ITCV_NOP=1"
run "ITCV_BN_BLOCKS=512"
run "ITCV_BN_BLOCKS=2048"
run "ITCV_BN_BLOCKS=4096"
run "ITCV_WGP_BLOCKS=240"
run "ITCV_WGP_BLOCKS=192"
run "ITCV_P2_BLOCKS=128"
run "ITCV_BAND_PERSIST=248"
run "ITCV_WGRAD_STREAM=0"
run "ITCV_NOP=1"
