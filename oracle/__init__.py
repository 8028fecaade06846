"""CPU oracle for the sky_embeddings hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker.  The product
(`sky_embeddings_amd`) never imports this package and fails loudly when its
HIP library is missing.

Pinning status:
  * everything restated from ``utils/mim_vit.py`` itself (input norm, NaN
    fill, random masking, token assembly, decoder un-shuffle, patchify, loss,
    init, optimiser wiring) and all of ``utils/similarity.py`` is pinned by
    golden vectors produced by running the reference in the build container
    (``tests/golden/make_golden.py``);
  * the transformer Block / PatchEmbed arithmetic lives in the un-vendored,
    un-pinned third-party ``timm`` package (reference call sites
    ``utils/mim_vit.py:6-8,206,231-233,276-278``): **parity unpinned** for
    that arithmetic -- it is restated from timm's published algorithm
    (pre-LN block, qkv-bias attention with softmax(q k^T / sqrt(hd)) v,
    exact-erf GELU MLP, Conv2d k=s=p patch embedding).
"""
