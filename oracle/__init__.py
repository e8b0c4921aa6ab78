"""CPU oracle for the Blurry-Edges hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU / numpy restatement of the reference algorithm (local_stage CNN, blurred-wedge
renderer + ridge colour solve, DfD depth solve, unfold/fold tiling), every function citing the
reference file:line it follows.  It is dtype-parametric (float32 reproduces the reference's
arithmetic order; float64 is the ground truth for the ill-conditioned stages).

Pinned: tests/golden/make_golden.py imported the real reference (/root/reference, PyTorch-CPU) in the build
container and wrote tests/golden/*.npz; tests/test_oracle_golden.py checks this oracle against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (blurry-edges_amd/) never does; it fails loudly when the HIP library is missing.
"""
