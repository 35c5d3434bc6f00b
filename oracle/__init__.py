"""CPU oracle for the audio -> face-mesh regression hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product package (``audio2face-pytorch_amd``) never imports this package
and raises if its HIP library is missing.

What it is: a functional, plain-torch (CPU, fp32) restatement of the reference
algorithm for the path named by BASELINE.json ``north_star``:

* ``oracle.voca``        <- /root/reference/src/model/voca.py:5-52
* ``oracle.audio2mesh``  <- /root/reference/src/model/audio2face.py:5-69
* ``oracle.faceformer``  <- /root/reference/src/model/faceformer.py:22-191
* ``oracle.wav2vec2``    <- /root/reference/src/model/wav2vec.py:76-189 and the
  HF ``transformers`` Wav2Vec2Model it subclasses (third party, un-vendored,
  un-pinned by the reference; container has transformers 5.15.0)
* ``oracle.losses``      <- /root/reference/src/loss/loss.py:4-55
* ``oracle.trainer``     <- /root/reference/src/model/lightning_model.py:119-125,145-161,209-213

Parity pinning: the reference ships no tests and no golden vectors,
so the oracle is pinned against outputs of the reference itself, produced in
the build container by ``tests/golden/gen_golden.py`` (which imports
``/root/reference`` and asserts oracle == reference before writing the
fixtures under ``tests/golden/*.npz``).  ``tests/test_oracle_golden.py``
re-checks the oracle against those committed fixtures everywhere (no access to
``/root/reference`` needed).  The wav2vec2 encoder arithmetic lives in HF
transformers; it is pinned the same way against the installed HF module with
formula-generated weights (no pretrained weights are available offline).

The reference is Python on torch; the oracle is therefore Python on torch CPU
ops (the same arithmetic library the reference's own CPU path uses), which is
what ``cpu_baseline.kind == "port"`` times in bench.py.
"""
