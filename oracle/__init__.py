"""CPU oracle for the ALG hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(choi403/ALG, /root/reference) used as the *checker* by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  Nothing in
the product package ``alg_amd`` imports it; the product path fails loudly when
the HIP library is missing instead of falling back to this code.

Pinning status ("Oracle"):
  * schedule + down_up + Hunyuan buckets: PINNED against the imported reference
    ``lp_utils`` (fixtures in tests/golden/, generator tests/golden/make_golden.py).
  * gaussian_blur: parity unpinned at the torchvision boundary (torchvision is
    not installed; restated from its published algorithm, cross-checked against
    an independent scipy formulation).
  * DiT forward / DDIM scheduler / RoPE helper: parity unpinned (the arithmetic
    lives in diffusers @ be2fb77, absent from /root/reference and this image).
"""
