"""Outcome of the REFERENCE's ``make_model`` argument checks over the whole argument grid -> ``make_model_grid.json``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_make_model_grid.py

``experiments/benchmark_utils.py`` cannot be imported here (hydra / omegaconf / pykeops are absent), but the validation part of
``make_model`` (:96-160, everything before ``with initialize(...)``) is plain Python over its arguments.  This script takes the
reference's own function from its file with ``ast`` -- the statements of ``make_model`` up to the first ``with`` -- compiles exactly
those statements and runs them for every combination of

    solver_type x ref_type x loss_type x integrator_type x model_type x time_type x force_base_zero_init x force_vp20 x force_vp_cosine

recording "ok" or the ValueError message.  The fixture is data: the grid axes, the distinct outcomes and one outcome index per
combination (row-major in the axis order above).  ``tests/test_make_model_contract.py`` holds the mirror to it.
"""
from __future__ import annotations

import ast
import itertools
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/experiments/benchmark_utils.py"


def reference_validator():
    tree = ast.parse(open(SRC).read())
    ns = {}
    for node in tree.body:  # the two name maps the checks read
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", None) in ("solver_types", "model_types"):
            exec(compile(ast.Module([node], []), SRC, "exec"), ns)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "make_model")
    cut = next(i for i, st in enumerate(fn.body) if isinstance(st, ast.With))
    fn.body = fn.body[:cut] + [ast.Return(ast.Constant("ok"))]
    ast.fix_missing_locations(fn)
    exec(compile(ast.Module([fn], []), SRC, "exec"), ns)
    return ns["make_model"], ns["solver_types"], ns["model_types"]


def main():
    validate, solver_types, model_types = reference_validator()
    axes = dict(solver_type=list(solver_types), ref_type=["default", "gaussian", "gmm", "nn"], loss_type=["kl", "lv"],
                integrator_type=["em", "ei", "ddpm_like"], model_type=list(model_types), time_type=["uniform", "snr"],
                force_base_zero_init=[False, True], force_vp20=[False, True], force_vp_cosine=[False, True])
    outcomes, index = [], []
    for combo in itertools.product(*axes.values()):
        kw = dict(zip(axes, combo))
        try:
            res = validate(solver_details={}, target_details={"name": "many_modes"}, training_details={}, **kw)
        except ValueError as e:
            res = "ValueError: " + str(e)
        if res not in outcomes:
            outcomes.append(res)
        index.append(outcomes.index(res))
    out = dict(source="experiments/benchmark_utils.py make_model, statements before `with initialize(...)`", axes=axes,
               outcomes=outcomes, index="".join(chr(ord("a") + i) for i in index))
    with open(os.path.join(HERE, "make_model_grid.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(len(index), "combinations;", {o: index.count(i) for i, o in enumerate(outcomes)})


if __name__ == "__main__":
    main()
