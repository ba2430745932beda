"""CPU oracle for the MAMDR hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a plain-numpy (fp32) restatement of the arithmetic on the
reference's hot path: the MLP tower forward/backward with
the Keras BCE loss, the TF1 Adam / SGD update, the 500-threshold AUC, the outer
Domain-Negotiation / Reptile / MAMDR parameter updates and the loops that drive
them.  Every function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product path (``mamdr_amd``) never imports it and has no CPU fallback.

PINNING STATUS
* Outer updates (oracle/outer.py): pinned bit-for-bit against vectors produced
  by the reference's own numpy methods (tests/golden/make_outer_goldens.py ran
  ``model_zoo.{mamdr,reptile,domain_negotiation,specific_base_model}`` from
  /root/reference with tensorflow/deepctr stubbed out).
* AUC (oracle/auc.py): pinned against the only known-answer vector in the
  reference, the docstring example at utils/auc.py:46-55 (AUC = 0.75).
* Inner step (oracle/tower.py: gather, MLP, BCE, Adam): **parity unpinned**.
  The arithmetic lives in tensorflow-gpu==1.12.0 and deepctr==0.9.0
  (requirements.txt:1,6), neither vendored nor installable here (python 3.10, no
  network), and the reference has no tests or golden vectors at that boundary.
  The restatement follows the published algorithms of those pinned versions
   anchored on the reference call sites
  (model_zoo/DeepCTR/deepctr.py:54-60,118-136).
"""
