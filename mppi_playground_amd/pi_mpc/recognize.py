"""Recognition of the reference EXAMPLES' own closures (round 5).

The reference's classic-control examples (example/pendulum.py:17-47, cartpole.py:17-81, mountaincar.py:17-55,
mujoco_cartpole.py:20-79) define `dynamics` as a TorchScript closure inside `main()` and the cost as a plain closure over a
module-level TorchScript `angle_normalize`; they carry no native tag, so `MPPI` used to run them on the generic path (the
reference's two T-step loops of tiny torch kernels).  This module recognises exactly those callables and lets the solver run the
corresponding fused model instead — the example files drop in unchanged AND at native throughput.

A pair is recognised only if BOTH tests pass:
  1. fingerprint: a sha256 of the callable's normalised source is listed in closure_fingerprints.json (hashes only, computed in
     the build container by scripts/closure_fingerprints.py from the reference's example files; no source text is kept).  Two
     forms are computed and either may match (round 6):
       "g2"  independent of the torch and Python version — TorchScript functions: the operator sequence of `inlined_graph`
             (node kinds with their input / output counts, nested blocks bracketed) plus the SORTED multiset of its constants —
             not the printer's text, whose layout, constant order and debug names change between releases; Python functions: a
             pre-order walk of the de-decorated AST (node classes, names, attributes, constants) — not `ast.dump`, whose fields
             change between Python versions; the TorchScript helpers a function names are appended in the same form;
       "v1"  round 5's form: `ScriptFunction.code` as printed by the torch the table was generated with / `ast.dump`;
  2. behaviour: on a seeded batch of probe states / actions (the model's working range and beyond its clamps) the callable and
     the shipped torch plugin of that model (envs/classic_control.py) agree to 1e-6 — on copies, so that a dynamics function that
     updates its argument in place (mountain car, SURVEY B-Q7) is compared on what it returns AND on what it leaves behind.
Anything else — an edited example, a callable that merely looks similar — fails a test and takes the generic path: never a
wrong model.  The two tests disagreeing is reported ONCE per process with `warnings.warn` (RecognitionWarning): a TorchScript
dynamics that behaves like a shipped model on the probes but is not in the table (an example scripted by a torch release whose
scripting frontend emits other operators: the solver then runs the generic path, 10-50x slower, and says so), and a callable
whose fingerprint is listed but whose values differ.  `MPPI(..., recognize_closures=False)` switches the recognition off;
`solver._recognition` says what happened."""
from __future__ import annotations

import ast
import hashlib
import inspect
import json
import os
import textwrap
import warnings
from typing import Callable, List, Optional, Set, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_TABLE = os.path.join(_HERE, "closure_fingerprints.json")
# model -> (dim_state, dim_control, probe ranges per state column, action range)
_MODELS = {
    "pendulum": (2, 1, [(-7.0, 7.0), (-10.0, 10.0)], (-3.0, 3.0)),
    "cartpole": (4, 1, [(-3.0, 3.0), (-3.0, 3.0), (-0.4, 0.4), (-3.0, 3.0)], (-4.0, 4.0)),
    "mountaincar": (2, 1, [(-1.4, 0.8), (-0.1, 0.1)], (-1.5, 1.5)),
    "mjcartpole": (4, 1, [(-1.5, 1.5), (-3.0, 3.0), (-0.4, 0.4), (-3.0, 3.0)], (-4.0, 4.0)),
}


class RecognitionWarning(UserWarning):
    """The source fingerprint and the behaviour of a callable disagree about its being one of the reference examples' closures."""


def _script_code(fn) -> Optional[str]:
    return fn.code if isinstance(fn, (torch.jit.ScriptFunction,)) else None


def _graph_signature(fn) -> Optional[str]:
    """Operator sequence + sorted constants of a TorchScript function's inlined graph (see the module docstring, form "g2")."""
    if not isinstance(fn, torch.jit.ScriptFunction):
        return None
    ops: List[str] = []
    consts: List[str] = []

    def walk(block):
        for n in block.nodes():
            kind = n.kind()
            if kind == "prim::Constant":
                try:
                    v = n.output().toIValue()
                except Exception:  # noqa: BLE001  (a constant without a Python value: keep its type)
                    v = str(n.output().type())
                consts.append(f"{type(v).__name__}:{v!r}")
                continue
            ops.append(f"{kind}/{len(list(n.inputs()))}>{len(list(n.outputs()))}")
            for blk in n.blocks():
                ops.append("(")
                walk(blk)
                ops.append(")")

    g = fn.inlined_graph
    walk(g)
    return "ops=" + " ".join(ops) + ";consts=" + " ".join(sorted(consts)) + f";args={len(list(g.inputs()))}"


def _ast_signature(node: ast.AST, helpers=()) -> str:
    """Pre-order walk of an AST: node classes with the attributes / constants they carry; identifiers (arguments, variables,
    globals, the function's own name) are numbered by first appearance, so that the signature is one of STRUCTURE — a
    transcription with other variable names has the same one; the behaviour test tells apart what structure cannot (form "g2")."""
    out: List[str] = []
    names: dict = {}

    def ident(name: str) -> str:
        return f"#{names.setdefault(name, len(names))}"

    def walk(n):
        label = type(n).__name__
        if isinstance(n, ast.Constant):
            label += f"={n.value!r}"
        elif isinstance(n, ast.Name):
            label += "=" + ident(n.id)
        elif isinstance(n, ast.Attribute):
            label += f"={n.attr}"
        elif isinstance(n, ast.arg):
            label += "=" + ident(n.arg)
        elif isinstance(n, ast.keyword):
            label += f"={n.arg}"
        out.append(label)
        if isinstance(n, (ast.Load, ast.Store, ast.Del)):
            return
        out.append("(")
        for field in ("args", "body", "value", "values", "func", "keywords", "left", "right", "op", "ops", "comparators", "operand",
                      "test", "orelse", "targets", "target", "elts", "slice", "lower", "upper", "step", "posonlyargs", "kwonlyargs",
                      "vararg", "kwarg", "defaults", "kw_defaults", "annotation", "returns", "iter", "items", "keys", "dims"):
            v = getattr(n, field, None)
            if v is None:
                continue
            for c in (v if isinstance(v, list) else [v]):
                if isinstance(c, ast.AST):
                    walk(c)
        out.append(")")

    walk(node)
    return " ".join(out) + "".join(f"\nhelper {ident(name)}=" + sig for name, sig in helpers)


def _python_def(fn):
    """(de-decorated, docstring-free FunctionDef, scope of names it can see) of a plain Python function, or None."""
    try:
        src = textwrap.dedent(inspect.getsource(fn))
        node = ast.parse(src).body[0]
    except (OSError, TypeError, SyntaxError, IndexError):
        return None
    if not isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
        return None
    node.decorator_list = []
    if node.body and isinstance(node.body[0], ast.Expr) and isinstance(getattr(node.body[0], "value", None), ast.Constant) \
            and isinstance(node.body[0].value.value, str):
        node.body = node.body[1:] or [ast.Pass()]
    scope = dict(getattr(fn, "__globals__", {}))
    if getattr(fn, "__closure__", None):
        scope.update({k: c.cell_contents for k, c in zip(fn.__code__.co_freevars, fn.__closure__)})
    return node, scope


def fingerprints(fn: Callable) -> Set[str]:
    """Every form of the callable's fingerprint (see the module docstring); empty when it has no retrievable source."""
    out: Set[str] = set()
    code = _script_code(fn)
    if code is not None:
        out.add(hashlib.sha256(("script:" + code).encode()).hexdigest())                       # v1
        out.add(hashlib.sha256(("g2:script:" + _graph_signature(fn)).encode()).hexdigest())    # g2
        return out
    got = _python_def(fn)
    if got is None:
        return out
    node, scope = got
    v1 = "py:" + ast.dump(node, include_attributes=False)
    helpers = []
    # the TorchScript helpers it calls by name (module globals or closure cells): v1 in name order with their printed text,
    # g2 in order of appearance with their graph signatures
    for name in sorted({n.id for n in ast.walk(node) if isinstance(n, ast.Name)}):
        helper = scope.get(name)
        if _script_code(helper) is not None:
            v1 += f"\n{name}=script:" + _script_code(helper)
    seen = []
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and n.id not in seen and _script_code(scope.get(n.id)) is not None:
            seen.append(n.id)
            helpers.append((n.id, _graph_signature(scope[n.id])))
    g2 = "g2:py:" + _ast_signature(node, helpers)
    out.add(hashlib.sha256(v1.encode()).hexdigest())
    out.add(hashlib.sha256(g2.encode()).hexdigest())
    return out


def fingerprint(fn: Callable) -> Optional[str]:
    """Round 5's single fingerprint (form "v1"), kept for scripts and tests that name it."""
    code = _script_code(fn)
    if code is not None:
        return hashlib.sha256(("script:" + code).encode()).hexdigest()
    got = _python_def(fn)
    if got is None:
        return None
    node, scope = got
    text = "py:" + ast.dump(node, include_attributes=False)
    for name in sorted({n.id for n in ast.walk(node) if isinstance(n, ast.Name)}):
        helper = _script_code(scope.get(name))
        if helper is not None:
            text += f"\n{name}=script:" + helper
    return hashlib.sha256(text.encode()).hexdigest()


def fingerprint_g2(fn: Callable) -> Optional[str]:
    """The version-independent form alone (what scripts/closure_fingerprints.py adds to the table)."""
    v1 = fingerprint(fn)
    rest = fingerprints(fn) - {v1}
    return next(iter(rest)) if rest else None


_table_cache = None


def _table():
    global _table_cache
    if _table_cache is None:
        try:
            _table_cache = json.load(open(_TABLE))
        except (OSError, ValueError):
            _table_cache = {}
    return _table_cache


def _probes(model: str, device, n: int = 256):
    ds, dc, ranges, (alo, ahi) = _MODELS[model]
    g = torch.Generator(device="cpu").manual_seed(20260927)
    s = torch.rand(n, ds, generator=g)
    for j, (lo, hi) in enumerate(ranges):
        s[:, j] = lo + (hi - lo) * s[:, j]
    a = alo + (ahi - alo) * torch.rand(n, dc, generator=g)
    a[:8] = torch.tensor([[alo], [ahi], [0.0], [-0.0], [1e-3], [-1e-3], [0.5 * alo], [0.5 * ahi]])[:, :dc]
    return s.to(device), a.to(device)


def _same(x: torch.Tensor, y: torch.Tensor) -> bool:
    """Equal to 1e-6 of the scale of EACH column (a state's columns differ by orders of magnitude: mountain car's velocity is
    ~0.07 against a position of ~1.4 — one global scale would check the small column 20x looser)."""
    if x.shape != y.shape or not bool(torch.isfinite(x).all()):
        return False
    x2, y2 = x.reshape(x.shape[0], -1), y.reshape(y.shape[0], -1)
    scale = y2.abs().amax(dim=0).clamp_min(1e-30)
    return bool(((x2 - y2).abs().amax(dim=0) <= 1e-6 * scale).all())


_warned: Set[str] = set()
last_report: dict = {}


def _warn_once(key: str, msg: str) -> None:
    if key not in _warned:
        _warned.add(key)
        warnings.warn(msg, RecognitionWarning, stacklevel=4)


def _behaves_like(model: str, dynamics: Callable, cost_func: Callable, device) -> bool:
    from envs import classic_control as cc

    nd, nc = getattr(cc, f"{model}_dynamics"), getattr(cc, f"{model}_cost")
    try:
        s, a = _probes(model, device)
        s1, s2 = s.clone(), s.clone()
        out1, out2 = dynamics(s1, a.clone()), nd(s2, a.clone())
        ok = _same(out1, out2) and _same(s1, s2)  # what it returns and what it leaves in its argument
        info = {"t": 0, "prev_state": s, "prev_action": a, "initial_state": s}
        return ok and _same(cost_func(s.clone(), a.clone(), dict(info)), nc(s.clone(), a.clone(), dict(info)))
    except Exception:  # noqa: BLE001  (a callable that cannot take the probes is not one of the examples' closures)
        return False


def match(dynamics: Callable, cost_func: Callable, dim_state: int, dim_control: int, device) -> Optional[Tuple[Callable, Callable]]:
    """-> the shipped (dynamics, cost) plugins of the model the two callables ARE, or None.  `last_report` (module attribute)
    says what was decided and why: {"model", "fingerprint", "behaviour", "torch", "table_torch"}."""
    global last_report
    table = _table()
    last_report = {"model": None, "fingerprint": False, "behaviour": None, "torch": torch.__version__,
                   "table_torch": table.get("_torch") if table else None}
    if not table:
        return None
    fd, fc = fingerprints(dynamics), fingerprints(cost_func)
    from envs import classic_control as cc

    for model, (ds, dc, _, _) in _MODELS.items():
        entry = table.get(model)
        if entry is None or (ds, dc) != (dim_state, dim_control):
            continue
        listed = bool(fd & set(entry.get("dynamics", ()))) and bool(fc & set(entry.get("cost", ())))
        if listed:
            ok = _behaves_like(model, dynamics, cost_func, device)
            last_report.update(model=model, fingerprint=True, behaviour=ok)
            if ok:
                return getattr(cc, f"{model}_dynamics"), getattr(cc, f"{model}_cost")
            _warn_once("listed:" + model,
                       f"pi_mpc.recognize: the callables' source fingerprint is that of the reference's {model} example, but their values "
                       "differ from the shipped plugin's on the probe batches: taking the generic path (opaque torch callables)")
            return None
    # not listed: a TorchScript dynamics (side-effect free by construction) that nevertheless BEHAVES like a shipped model gets a word
    if isinstance(dynamics, torch.jit.ScriptFunction):
        for model, (ds, dc, _, _) in _MODELS.items():
            if (ds, dc) == (dim_state, dim_control) and table.get(model) and _behaves_like(model, dynamics, cost_func, device):
                last_report.update(model=model, fingerprint=False, behaviour=True)
                _warn_once("unlisted:" + model,
                           f"pi_mpc.recognize: these callables behave exactly like the shipped {model} plugin on the probe batches, but their "
                           f"source fingerprint is not in closure_fingerprints.json (running torch {torch.__version__}, table generated "
                           f"with torch {table.get('_torch')}): the solver takes the GENERIC path — the reference's two T-step loops "
                           "of small torch kernels, 10-50x slower than the fused model.  Pass the tagged plugins of "
                           f"envs.classic_control ({model}_dynamics / {model}_cost) for native throughput, or "
                           "recognize_closures=False to silence this")
                break
    return None
