"""CPU oracle for the myrtle-vision ViT hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and there only as the checker.  The product
package (``myrtle-vision_amd/myrtle_vision``) never imports it and fails loudly
when the HIP library is missing.

Parity status
-------------
* fp32 ViT forward/backward (classification + segmentation decoders): PINNED by
  golden vectors generated from the reference's own ``ViT`` imported in the
  build container (``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``).
* qtorch fake-quant rounding (``float_quantize``/``fixed_point_quantize``):
  PARITY UNPINNED -- qtorch==0.3.0 is not in the container and the reference
  holds no test vectors for it; ``oracle/quant_oracle.py`` restates the
  published algorithm.  The *placement* of the quantisers is pinned (golden
  site list + golden logits produced by the reference plumbing calling the
  restated quantiser).
* timm AdamW param groups / cosine schedule: PARITY UNPINNED (timm absent,
  no reference tests); restated in ``oracle/optim_oracle.py``.
"""
