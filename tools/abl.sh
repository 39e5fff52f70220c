for a in 0 8 23 31 1 3; do echo -n "ablate=$a: "; ITCV_ABLATE=$a python tools/bench_layers.py 2>&1 | grep "128-> 128@ 32" | grep -o "bf16x3 *[0-9.]* ms *[0-9.]* TFeq"; done
