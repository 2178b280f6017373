"""Container-only: golden vectors for the volume-level ACDC pipeline (tests/golden/acdc.npz).

  * AC17Data.read_files (/root/reference/data/ac17_dataloader.py:80-98), compiled from the file where it lies with `ast` and run from the
    reference's own directory so that it reads ITS data/data_series.txt: the 5-fold train / validation membership for every k_split.
Only inputs / outputs are stored (the series list is data: 200 "<patient> <frame>" pairs).  The in-plane re-scaling
(skimage.transform.rescale) cannot be run here -- skimage is not installed -- and stays an unpinned restatement with KATs in tests/test_acdc.py.
"""
import ast
import os
import sys

import numpy as np

REF = "/root/reference"


def main():
    sys.dont_write_bytecode = True
    src = open(os.path.join(REF, "data", "ac17_dataloader.py")).read()
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AC17Data"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "read_files"][0]
    ns = {"os": os}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "ac17_dataloader.py", "exec"), ns)
    read_files = ns["read_files"]

    class Self:
        pass
    out = {}
    series = [tuple(int(t) for t in l.split()[:2]) for l in open(os.path.join(REF, "data", "data_series.txt")) if l.strip()]
    out["series"] = np.array(series, np.int32)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        for k_split in (1, 2, 3, 4, 5):
            for split in ("train", "val"):
                s = Self(); s.k, s.split_len, s.k_split, s.split = 5, int(200 / 5), k_split, split
                out["fold%d.%s" % (k_split, split)] = np.array(read_files(s), np.int32)
    finally:
        os.chdir(cwd)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "acdc.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
