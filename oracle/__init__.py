"""CPU oracle for the Stable Diffusion DDPM train_step hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package.  The product (``stable_diffusion_training_amd``)
never imports it and fails loudly when its HIP library is missing.

What it is: a plain NumPy / PyTorch-CPU fp32 restatement of the algorithm the
reference executes in ``training_utils.py:504-762`` (``train_step``) together
with the third-party arithmetic that function reaches (diffusers 0.21.4 Flax
UNet / VAE encoder, Flax CLIP text model, optax clip + Lion).  Every function
cites the reference file:line (or the third-party module) it follows.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 and its JAX/Flax stack is not installed here, so nothing
the reference itself produced anchors this restatement.  The pins available
are (i) the hand-derived known-answer values of SURVEY.md §8(c), checked in
``tests/test_oracle_kat.py``, (ii) exact parameter-count matches against the
published model sizes, and (iii) internal consistency (analytic gradients vs
autograd, numpy vs torch restatements).
"""
