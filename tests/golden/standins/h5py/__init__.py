"""Stand-in for h5py (absent from the build image), used ONLY by tests/golden/make_golden_data.py to run the reference's
dataset classes (datamodule/dataset.py, dataset_2d.py) over an in-memory "file": ``File(path, 'r')`` returns the nested
mapping registered for ``path`` - groups are dicts, datasets are numpy arrays (both support the [] access the reference
uses).  No arithmetic lives here."""
_FILES = {}


def register(path, tree):
    _FILES[path] = tree


class File(dict):
    def __init__(self, path, mode="r"):
        super().__init__(_FILES[path])

    def close(self):
        pass
