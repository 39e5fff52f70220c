# Operand-ablation timings of the planes kernels: needs the diagnostic build of the library (-DITCV_DIAG), which is
# the only build in which ITCV_ABLATE has any effect.
make -C intro-tc-vae_amd/csrc diag -j6 >/dev/null || exit 1
export ITCV_LIB=$PWD/intro-tc-vae_amd/lib/libitcv_hip_diag.so
for a in ${ABL:-0 1 4 5 8 13}; do echo -n "ablate=$a: "; ITCV_ABLATE=$a python tools/bench_planes.py 2>&1 | grep "bf16x3" | grep "128-> 128@ 32\|512-> 512@  8\|512-> 256@ 16" | grep -o "planes *[0-9.]* us" | tr '\n' ' '; echo; done
