"""Golden vectors of the reference's ``utils.repeat_expand_2d`` (utils.py:482-496) -> golden_v3.npz ``g11.*``.

BUILD CONTAINER ONLY (reads /root/reference).  ``utils.py`` as a module does not import here (librosa / fairseq are absent),
so the generator takes the ONE function out of the reference file with ``ast`` and executes the reference's own code; what is
committed are inputs-by-seed and outputs (data), never the source.  Run: python tests/golden/make_golden_v3.py
"""
import ast
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
CASES = [(3, 50, 94), (2, 469, 938), (2, 499, 938), (2, 141, 282), (2, 100, 100), (4, 7, 5), (1, 1, 9), (3, 333, 1000), (2, 1407, 2813)]


def reference_function():
    src = open(os.path.join(REF, "utils.py")).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "repeat_expand_2d")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), os.path.join(REF, "utils.py"), "exec"), ns)
    return ns["repeat_expand_2d"]


def main():
    from ns2vc_amd.audio import repeat_expand_2d
    from ns2vc_amd.weights import hash_normal
    ref = reference_function()
    out, report = {}, {}
    for h, s, t in CASES:
        x = torch.from_numpy(hash_normal(f"g11.{h}.{s}.{t}", (h, s)))
        y = ref(x, t)
        mine = repeat_expand_2d(x, t)
        assert torch.equal(mine, y), (h, s, t)
        out[f"g11.{h}_{s}_{t}.y"] = y.numpy()
        report[f"g11.{h}_{s}_{t}"] = "bit-identical"
    np.savez_compressed(os.path.join(HERE, "golden_v3.npz"), **out)
    json.dump({"cases": CASES, "report": report}, open(os.path.join(HERE, "golden_v3_report.json"), "w"), indent=1)
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
