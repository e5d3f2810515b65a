"""Stand-in torch_geometric (golden generation only); arithmetic lives in oracle/pyg_semantics.py."""
