"""Generate tests/golden/random_sampling.npz: outputs of the REFERENCE's own host sub-sampling

    random_sampling      /root/reference/utils/pc_util.py:36-44

on seeded inputs, for the three regimes it has (fewer rows requested than present: without replacement; more: with
replacement; replacement forced).  DATA only.  `utils/pc_util.py` cannot be imported as a module here (its import block
pulls in matplotlib / trimesh, which this image does not have and the function does not use); no stand-in is made: the
function definition is taken out of the reference file IN PLACE with `ast` and executed against the real numpy.

    python tests/golden/make_golden_random_sampling.py
"""
import ast
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("OMNIPQ_REFERENCE", "/root/reference")

import numpy as np  # noqa: E402

CASES = [("fewer", 5, 50000, 40000, None), ("more", 6, 30000, 40000, None), ("forced", 7, 5000, 1000, True),
         ("tiny", 8, 7, 7, None)]


def take(path, names, ns):
    tree = ast.parse(open(path).read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)


def cloud(seed, n):
    return np.random.RandomState(1000 + seed).rand(n, 6).astype(np.float32)


def main():
    ns = {"np": np}
    take(os.path.join(REF, "utils", "pc_util.py"), ["random_sampling"], ns)
    out = {}
    for name, seed, n, k, replace in CASES:
        pc = cloud(seed, n)
        np.random.seed(seed)
        sub, choices = ns["random_sampling"](pc, k, replace=replace, return_choices=True)
        out[f"{name}.args"] = np.array([seed, n, k, -1 if replace is None else int(replace)], np.int64)
        out[f"{name}.choices"] = choices.astype(np.int64)
        out[f"{name}.checksum"] = sub.astype(np.float64).sum(0)
    np.savez_compressed(os.path.join(HERE, "random_sampling.npz"), **out)
    print("wrote random_sampling.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
