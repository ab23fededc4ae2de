"""oracle/ -- CPU restatement of the reference's Karras-EDM sampling path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from here, and only as the checker / reported
baseline.  ``diffsci_amd`` never imports it; the product path raises when the
HIP library is missing instead of falling back to this code.

What it is: our own functional (state_dict-driven) torch-CPU restatement of
  * the sigma grid / preconditioner scalars / ODE-SDE right-hand side /
    Euler, Heun, Euler-Maruyama and Karras (sigma-churn) steppers
    (reference: diffsci/models/karras/{schedulers,integrators,preconditioners,
    schedulingfunctions,karrasmodule}.py), module ``oracle.karras_ref``;
  * the PUNetG score network and the toy MLP
    (reference: diffsci/models/nets/{punetg,commonlayers,attention,mlp}.py),
    modules ``oracle.punetg_ref`` / ``oracle.mlp_ref``.
Every function cites the reference file:line it follows.

Pinned: the restatement is checked bit-for-bit (fp32 and fp64) against golden
vectors produced by importing the real reference in the build container
(``oracle/tools/make_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.  The convolution / GEMM / GroupNorm / softmax
arithmetic inside the network is torch's (a third-party dependency of the
reference, pinned only as "torch" in its requirements.txt); the oracle calls the
same torch CPU operators, so it reproduces the reference's numbers exactly on
the same host ISA.
"""
