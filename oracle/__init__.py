"""CPU oracle for the LTX-2 denoise + VAE-decode hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ltx-2-mlx_amd/`` (the product) may
import this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

The oracle is a plain PyTorch fp32 restatement of the arithmetic of the
reference (Acelogic/LTX-2-MLX, Python on Apple MLX).  Every function cites
the reference file:line it follows.  The reference cannot be imported in this
environment (``import mlx`` fails on Linux) and ships no
golden tensors for this path, so:

  * closed-form known answers the reference's own tests assert (sigma tables,
    post_process_latent, timesteps_from_mask, latent-shape arithmetic,
    patchify round trips) are re-asserted against this oracle in
    ``tests/test_oracle_known_answers.py``;
  * ``tools/pin_oracle_against_reference.py`` executes the reference's own
    source files from /root/reference through a throw-away ``mlx`` -> torch
    shim (leaf ops only) and writes small golden vectors to ``tests/golden``;
    ``tests/test_oracle_golden.py`` checks the oracle against them;
  * parity against the upstream Lightricks PyTorch code is UNPINNED (that
    repository is an un-vendored submodule, absent from /root/reference).
"""
