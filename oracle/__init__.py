"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the SM3 pre-training hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker / the timed
CPU baseline.  The product path (``skin-sm3_amd/``) never imports it and fails loudly when
its HIP library is missing.

Contents
--------
sm3_oracle.py   torch-CPU functional restatement of the reference hot path (fp32 / fp64):
                ResNet-50 encoder, projector, NT-Xent logits, loss composition, AdamW.
                Pinned against fixtures generated from the reference itself
                (``tests/golden/*.npz`` written by ``gen_golden.py``).
procedural.py   seeded, platform-independent weight / input generators shared by the golden
                generator and the tests (weights are never stored, only regenerated).
ref_stub.py     torchvision *metadata* stub that lets the reference's Python import in the
                build container (the reference never travels to the GPU box).
gen_golden.py   imports /root/reference (build container only) and writes the fixtures.
"""
