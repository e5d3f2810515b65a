"""Shadow of the reference's ``models`` package: put ``<repo>/shim`` in front of the reference checkout on
``PYTHONPATH`` and ``from models.factory import FACTORY`` (run.py:8, tune.py:10) resolves to the MI355X implementation."""
