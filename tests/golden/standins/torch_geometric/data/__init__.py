class Data:
    def __init__(self, x=None, edge_index=None, **kw):
        self.x = x
        self.edge_index = edge_index
        for k, v in kw.items():
            setattr(self, k, v)
