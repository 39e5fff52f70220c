"""hipvae: host-side plumbing between the reference-shaped Python surface (ops / models /
solvers) and libitcv_hip.so.  abi = ctypes binding, functional = autograd Functions,
flat = flat parameter/gradient/Adam buffers, ddp = data-parallel context (RCCL)."""
