"""CPU oracle for the Soft-Intro beta-TC-VAE training path.  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU restatement (fp32, fp64-capable) of the reference
algorithm (meffmadd/intro-tc-vae: ops.py, models.py, solvers/*.py).  It exists to CHECK
the hand-written HIP path and to time a CPU baseline; it is never the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
    may import it;
  * nothing under ``intro-tc-vae_amd/`` imports it, and the product raises if the HIP
    extension is missing instead of falling back to anything here.

Parity status: PINNED.  Every function here is checked against golden vectors that were
produced by importing the unmodified reference in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).

Modules
  latent_math  ops.py restatement (densities, MSS/MWS, TC, KL, reparameterise, recon loss)
  network      models.py restatement as a functional layer program over a state dict
  steps        solvers/{vae,tc,intro,intro_tc}.py restatement (train_step) + Adam + clip
"""
