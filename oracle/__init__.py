"""CPU oracle for the MAgNet rollout hot path.  TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a plain torch-CPU restatement of the reference
algorithm (jaggbow/magnet, mounted read-only at /root/reference while this repo
was built).  It exists to *check* the HIP path and to be timed as the
``cpu_baseline`` leg of ``bench.py``.  Only ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s cpu_baseline may import it; the product package
``magnet_amd`` never does and fails loudly when its HIP library is missing.

Parity status
-------------
* Pinned: every ``forward`` / ``validation_step`` restated here is checked
  (tests/test_oracle_golden.py) against golden vectors produced by importing the
  reference's own classes (tests/golden/make_golden.py, run in the build
  container where /root/reference exists).
* UNPINNED: the reference calls torch_geometric 2.0.3 (MessagePassing,
  InstanceNorm) and torch_cluster (radius_graph, knn), neither of which is
  vendored in the reference nor installable offline.  Their arithmetic is
  restated in ``oracle/pyg_semantics.py`` from the libraries' published
  algorithms; the golden generator imports the reference on top of that same
  restatement.  No reference test or fixture pins those semantics
  ("parity unpinned" at the PyG / torch_cluster boundary).
"""
