"""Column history behind `train.csv` (the loss-vs-wallclock artefact): same interface and file
format as the reference's `History` (reference src/odil/history.py:9-123) so that callbacks written
for it (`history.append(key, value)`) and the consumers of `train.csv` keep working.

Rows are typed columns; a column that appears late is back-filled with zeros of its type, a value
of None repeats a zero of the column's type; `write()` commits the row and appends the pending
rows to the CSV (the header is fixed at the first write, after `warmup` rows were collected)."""

import pickle

import numpy as np


class History:
    def __init__(self, csvpath=None, warmup=0):
        self.data = dict()
        self.count = 0
        self.warmup = warmup
        self.csvcount = 0
        self.csvpath = csvpath
        self.csvkeys = None
        self.csvfile = open(csvpath, "w") if csvpath is not None else None

    @staticmethod
    def _zero_of(value):
        if value is None:
            return None
        if isinstance(value, (float, np.floating)):
            return 0.0
        if isinstance(value, (int, np.integer)):
            return 0
        raise ValueError("Unknown type: " + str(type(value)))

    def append(self, key, value=None):
        if isinstance(value, np.ndarray):
            assert value.ndim == 0 or value.shape == (1,), "Expected a scalar, got shape " + str(value.shape)
            value = value.item()
        if hasattr(value, "detach"):  # 0-d device tensor
            value = value.detach().cpu().item()
        assert value is None or isinstance(value, (int, float, str, np.integer, np.floating)), (
            "Unexpected type: " + str(type(value)))
        column = self.data.get(key)
        if column is None:
            assert value is not None, "First value of column '{}' is None".format(key)
            column = self.data[key] = [self._zero_of(value)] * self.count
        if value is None:
            assert column, "Expected non-empty column " + key
            value = self._zero_of(column[-1])
        column.append(value)

    def append_dict(self, newdict):
        for k, v in newdict.items():
            self.append(k, v)

    def get(self, key, default=None):
        return self.data.get(key, default)

    def commit(self):
        longest = max(len(v) for v in self.data.values())
        short = [k for k, v in self.data.items() if len(v) < longest]
        if short:
            raise RuntimeError("Missing values for columns: " + ",".join(short) + ",")
        self.count += 1

    def write(self, nocommit=False):
        if not nocommit:
            self.commit()
        if self.count <= self.warmup or self.csvfile is None:
            return
        if self.csvkeys is not None and len(self.data) != len(self.csvkeys):
            raise RuntimeError("Unexpected keys in history: {:}".format(sorted(set(self.data) - set(self.csvkeys))))
        if self.csvcount == 0:
            self.csvkeys = list(self.data)
            self.csvfile.write(",".join(self.csvkeys) + "\n")
        for row in range(self.csvcount, self.count):
            self.csvfile.write(",".join(str(self.data[k][row]) for k in self.data) + "\n")
        self.csvcount = self.count
        self.csvfile.flush()

    def save(self, path):
        with open(path, "wb") as f:
            pickle.dump(self.data, f)

    def load(self, path):
        with open(path, "rb") as f:
            self.data = pickle.load(f)
        self.csvkeys = list(self.data)
        self.count = len(next(iter(self.data.values())))
        self.write(nocommit=True)

    def close(self):
        if self.csvfile:
            self.csvfile.close()
