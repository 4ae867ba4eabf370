"""Shared test helpers: golden-fixture loading and oracle problem construction."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

# case name -> (model, ctor kwargs as used by tests/golden/make_golden.py)
CASES = {
    "pendulum_T50_N1000_essps": dict(model="pendulum", T=50, N=1000, lambda_="ESSPS"),
    "pendulum_T15_N256_fixed": dict(model="pendulum", T=15, N=256, lambda_=1.0),
    "pendulum_T15_N256_lbps": dict(model="pendulum", T=15, N=256, lambda_="LBPS"),
    "pendulum_T15_N256_mpo": dict(model="pendulum", T=15, N=256, lambda_="MPO"),
    "pendulum_T15_N200_explore": dict(model="pendulum", T=15, N=200, lambda_=0.5, exploration=0.25),
    "cartpole_T64_N1024_essps_sg": dict(model="cartpole", T=64, N=1024, lambda_="ESSPS", use_sg_filter=True),
    "cartpole_T10_N100_fixed": dict(model="cartpole", T=10, N=100, lambda_=0.001),
    "mountaincar_T100_N256_fixed": dict(model="mountaincar", T=100, N=256, lambda_=0.1),
    "nav2d_T50_N512_essps": dict(model="nav2d", T=50, N=512, lambda_="ESSPS"),
    "nav2d_T30_N256_fixed_explore": dict(model="nav2d", T=30, N=256, lambda_=1.0, exploration=0.25),
    "racing_T50_N512_fixed": dict(model="racing", T=50, N=512, lambda_=1.0),
    "racing_T25_N256_fixed": dict(model="racing", T=25, N=256, lambda_=1.0),
    "mjcartpole_T50_N256_fixed": dict(model="mjcartpole", T=50, N=256, lambda_=1.0),
    "goalzone_T30_N256_fixed": dict(model="goalzone", T=30, N=256, lambda_=1.0),
    # round 2: dense softmax at N = 4096, racing / nav2d with SG + exploration, the other
    # temperature rules on nav2d, ESSPS running into lambda_max, a posterior draw between two solves
    "racing_T25_N4096_dense": dict(model="racing", T=25, N=4096, lambda_=500.0),
    "racing_T25_N512_explore_sg": dict(model="racing", T=25, N=512, lambda_=200.0, exploration=0.25,
                                       use_sg_filter=True),
    "racing_T25_N1024_essps": dict(model="racing", T=25, N=1024, lambda_="ESSPS"),
    "nav2d_T30_N4096_essps": dict(model="nav2d", T=30, N=4096, lambda_="ESSPS"),
    "nav2d_T30_N512_lbps": dict(model="nav2d", T=30, N=512, lambda_="LBPS"),
    "nav2d_T30_N512_mpo": dict(model="nav2d", T=30, N=512, lambda_="MPO"),
    "nav2d_T30_N512_sg": dict(model="nav2d", T=30, N=512, lambda_=5.0, use_sg_filter=True, sg_window_size=7,
                              sg_poly_order=2),
    "nav2d_T20_N256_posterior": dict(model="nav2d", T=20, N=256, lambda_=5.0),
    # round 5: the other temperature rules at the examples' sample count on nav2d AND racing; ESSPS at both end-point rules
    # (mppi.py:361-364).  Noise by seed (see Fixture).
    "nav2d_T30_N4096_lbps": dict(model="nav2d", T=30, N=4096, lambda_="LBPS"),
    "nav2d_T30_N4096_mpo": dict(model="nav2d", T=30, N=4096, lambda_="MPO"),
    "racing_T25_N4096_lbps": dict(model="racing", T=25, N=4096, lambda_="LBPS"),
    "racing_T25_N4096_mpo": dict(model="racing", T=25, N=4096, lambda_="MPO"),
    "nav2d_T30_N512_essps_at_min": dict(model="nav2d", T=30, N=512, lambda_="ESSPS", lambda_min=40.0, lambda_max=100.0),
    "nav2d_T30_N512_essps_at_max": dict(model="nav2d", T=30, N=512, lambda_="ESSPS", lambda_max=0.5),
}
# ctor kwargs a case may carry
SOLVER_KW = ("exploration", "use_sg_filter", "sg_window_size", "sg_poly_order", "lambda_min", "lambda_max")


def sg_coeffs(cfg):
    from pi_mpc import _host

    return _host.savitzky_golay_coeffs(cfg.get("sg_window_size", 5), cfg.get("sg_poly_order", 3))

MODEL_CFG = {
    "pendulum": dict(u_min=[-2.0], u_max=[2.0], sigmas=[1.0]),
    "cartpole": dict(u_min=[-3.0], u_max=[3.0], sigmas=[1.0]),
    "mountaincar": dict(u_min=[-1.0], u_max=[1.0], sigmas=[1.0]),
    "nav2d": dict(u_min=[0.0, -1.0], u_max=[2.0, 1.0], sigmas=[0.5, 0.5]),
    "racing": dict(u_min=[-2.0, -0.25], u_max=[2.0, 0.25], sigmas=[0.5, 0.1]),
    "mjcartpole": dict(u_min=[-3.0], u_max=[3.0], sigmas=[1.0]),
    "goalzone": dict(u_min=[-1.0, -1.0], u_max=[1.0, 1.0], sigmas=[0.5, 0.5]),
}


class Fixture:
    """An .npz fixture.  Cases recorded with `eps_by_seed` (round 5) do not store the [N,T,dc] noise blocks: they are
    torch's CPU stream from the seed (MultivariateNormal.rsample == randn * sigma, SURVEY B-Q1) — the constructor's draw
    first, then one block per solve (mppi.py:146-148,261-263).  `eps_k` / `ctor_eps` are drawn again with torch on first
    use (what the product's noise_source="torch_cpu" does) and verified bit for bit against the float64 checksum and the
    first / last rows the fixture keeps.  (The oracle's restatement of the stream agrees to 2e-6 only — libm against
    torch's vectorised log / sin / cos — which is not enough for an end-to-end comparison with the reference.)"""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self._eps = None
        self.by_seed = "eps_by_seed" in self._z.files
        self.files = list(self._z.files)
        if self.by_seed:
            self.files += ["ctor_eps"] + [f"eps_{k}" for k in range(int(self._z["K"]))]

    def _regen(self):
        if self._eps is None:
            z = self._z
            cfg = CASES[self._name()] if self._name() in CASES else None
            N, T = (cfg["N"], cfg["T"]) if cfg else (int(z["N"]), int(z["T"]))
            sig = np.asarray(z["sigmas"], np.float32)
            import torch  # the reference's own sampler (a third-party dependency of the reference, SURVEY B-Q1)

            gen = torch.Generator(device="cpu").manual_seed(int(z["eps_by_seed"]))
            blocks = []
            for k in range(-1, int(z["K"])):  # block -1 is the constructor's draw
                e = (torch.randn(N, T, len(sig), generator=gen, dtype=torch.float32) * torch.from_numpy(sig)).numpy()
                if k >= 0:  # bit for bit, or this machine's torch does not draw the reference's stream
                    assert np.array_equal(e[:2], z[f"eps_head_{k}"]) and np.array_equal(e[-1:], z[f"eps_tail_{k}"])
                    assert float(e.astype(np.float64).sum()) == float(z[f"eps_sum64_{k}"])
                blocks.append(e)
            self._eps = blocks
        return self._eps

    def _name(self):
        return os.path.basename(self._z.fid.name)[:-4] if getattr(self._z, "fid", None) is not None else ""

    def __getitem__(self, key):
        if self.by_seed and (key == "ctor_eps" or key.startswith("eps_") and key[4:].isdigit()):
            return self._regen()[0 if key == "ctor_eps" else int(key[4:]) + 1]
        return self._z[key]

    def __contains__(self, key):
        return key in self.files


def load(name):
    f = Fixture(name)
    f._name = lambda: name
    return f


def unpack_bits(bits, shape):
    n = int(shape[0]) * int(shape[1])
    return np.unpackbits(bits)[:n].reshape(int(shape[0]), int(shape[1])).astype(np.uint8)


_env_cache = {}


def racing_env_fixture():
    if "racing" not in _env_cache:
        e = load("racing_env")
        shape = e["map_shape"]
        _env_cache["racing"] = dict(
            obst=unpack_bits(e["obst_bits"], shape), lane=unpack_bits(e["lane_bits"], shape),
            cell=float(e["cell_size"]), origin=e["origin"].astype(np.float64),
            center_path=e["center_path"], center_path_f64=e["center_path_f64"], circles=e["circles"],
            start_state=e["start_state"], x_lim=e["x_lim"], y_lim=e["y_lim"],
            lane_width=float(e["lane_width"]))
    return _env_cache["racing"]


def nav2d_env_fixture():
    if "nav2d" not in _env_cache:
        e = load("nav2d_env")
        _env_cache["nav2d"] = dict(
            map=unpack_bits(e["map_bits"], e["map_shape"]), cell=float(e["cell_size"]),
            origin=e["origin"].astype(np.float64), circles=e["circles"], rects=e["rects"],
            goal=e["goal"], start_state=e["start_state"], x_lim=e["x_lim"], y_lim=e["y_lim"])
    return _env_cache["nav2d"]


def goalzone_env_fixture():
    if "goalzone" not in _env_cache:
        e = load("goalzone_env")
        _env_cache["goalzone"] = dict(goal=e["goal"], center=e["center"], radius=float(e["radius"]), x0=e["x0"])
    return _env_cache["goalzone"]


def oracle_problem(model, N, T, exploration=0.0, ref_path=None):
    cfg = MODEL_CFG[model]
    params, maps = (), ()
    if model == "racing":
        e = racing_env_fixture()
        params = orc.racing_params()
        maps = [(e["obst"], e["cell"], e["origin"]), (e["lane"], e["cell"], e["origin"])]
    elif model == "nav2d":
        e = nav2d_env_fixture()
        params = orc.nav2d_params()
        maps = [(e["map"], e["cell"], e["origin"])]
    elif model == "goalzone":
        e = goalzone_env_fixture()
        params = orc.goalzone_params(np.float32(e["goal"]), np.float32(e["center"]), e["radius"])
    return orc.Problem(model, N, T, cfg["u_min"], cfg["u_max"], exploration=exploration, params=params,
                       maps=maps, ref_path=ref_path)


# ---- the reference's own measured sensitivity (tests/golden/make_golden.py: BAND_VARIANTS; 256 probes per solve)
# the probes by row of every band array (make_golden.py: BAND_VARIANTS): what a coinciding probe is called in the report
N_ULP, N_STAGE_ULP = 160, 92
BAND_VARIANTS = (tuple(f"ulp_{i}" for i in range(N_ULP)) + ("f64sum", "seqsum", "revsum", "pairsum")
                 + tuple(f"stage_ulp_{i}" for i in range(N_STAGE_ULP)))


class Band(float):
    """A band: the sample MAXIMUM over the probes (its float value, what the tests compare with) that also carries the probes
    themselves — every probe's error, sorted, with the probe's name — so that a check can say WHERE in the reference's own
    spread the device's error falls (`rank`) and whether it IS one of the probes bit for bit (`coincides`: the device then
    computes exactly that equally valid fp32 evaluation of the costs — e.g. "f64sum", the exactly rounded sum — and its
    distance to the reference's fixture is that probe's, to the last digit)."""

    def __new__(cls, values, names=None):
        v = np.asarray(values, np.float64).ravel()
        b = super().__new__(cls, float(v.max()) if v.size else 0.0)
        b.p99 = float(np.percentile(v, 99)) if v.size else 0.0
        b.probes = int(v.size)
        order = np.argsort(v, kind="stable")
        b.sorted = v[order]
        nm = list(names) if names is not None else [f"probe_{i}" for i in range(v.size)]
        b.names = [nm[i] for i in order]
        return b

    def rank(self, err: float) -> int:
        """How many of the reference's probes moved its output by no more than `err` (0 .. probes)."""
        return int(np.searchsorted(self.sorted, err, side="right"))

    def coincides(self, err: float, limit: int = 4):
        """Names of the probes whose error equals `err` (both are max|a - b| / max|b| in float64 of fp32 arrays against the same
        fixture: equal outputs give equal numbers)."""
        if not self.probes or err == 0.0:
            return []
        hit = np.nonzero(np.abs(self.sorted - err) <= 1e-12 * max(err, 1e-300))[0]
        return [self.names[i] for i in hit[:limit]]

    @staticmethod
    def merge(*bands):
        vals = np.concatenate([b.sorted for b in bands]) if bands else np.zeros(0)
        names = [n for b in bands for n in b.names]
        return Band(vals, names)


def _probe_names(g, key, prefix):
    sel = g[key] if key in g.files else None  # (full-size fixtures record a subset of the probes)
    names = BAND_VARIANTS if sel is None else [BAND_VARIANTS[int(i)] for i in sel]
    return [prefix + n for n in names]


def band_fixed(g, k):
    """(action, state): how far the REFERENCE's action_seq / state_seq of solve k move when its total costs are replaced
    by equally valid fp32 evaluations of the same sums (1-ulp changes, other summation orders), inputs and temperature
    held fixed.  Maximum over the recorded probes."""
    b = g[f"band_fixed_{k}"]
    nm = _probe_names(g, "band_variants_fixed", "fixed:")
    return Band(b[:, 0], nm), Band(b[:, 1], nm)


def band_rule(g, k):
    """(action, state, lambda) of the same probes with the reference's temperature rule re-run, or None for a fixed temperature."""
    if f"band_rule_{k}" not in g.files:
        return None
    b = g[f"band_rule_{k}"]
    nm = _probe_names(g, "band_variants_fixed", "rule:")
    return Band(b[:, 0], nm), Band(b[:, 1], nm), Band(b[:, 2], nm)


def band_rule_lambda(g, k):
    """The same probes with the reference's temperature rule re-run: relative spread of the temperature of solve k."""
    r = band_rule(g, k)
    return r[2] if r is not None else Band([0.0])


def band_closed_loop(g, k):
    """dict(x0, action, state, lam): the reference's whole K-solve closed loop re-run per probe (states, warm start, SG
    history and the rule's memory evolve on their own), relative distance of solve k to the unperturbed loop."""
    b = g["band_closed_loop"][k]
    nm = _probe_names(g, "band_variants_closed", "closed:")
    return dict(x0=Band(b[:, 0], nm), action=Band(b[:, 1], nm), state=Band(b[:, 2], nm), lam=Band(b[:, 3], nm))


def rel_err(a, b):
    """max |a-b| relative to max |b| (the tolerance convention of SURVEY Appendix D)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def lbps_objective64(costs, lam, delta=0.01):
    """The LBPS objective (src/pi_mpc/mppi.py:534-557) in float64 — the yardstick for a temperature found on the
    reference's fp32, noise-flat objective: two minimisers are "the same" when this differs by a few fp32 ulps."""
    c = np.asarray(costs, np.float64)
    x = -c / lam
    e = np.exp(x - x.max())
    w = e / e.sum()
    ess = 1.0 / np.sum(w * w)
    return float(-(-np.sum(w * c) - (c.max() - c.min()) * np.sqrt((1 - delta) / delta) / np.sqrt(ess)))


def same_lbps_minimum(costs, lam, lam_ref, delta=0.01, tol=2e-2):
    """LBPS temperatures agree: within 1e-3 relative, or — where the fp32 objective is flat to its own rounding noise
    (nav2d: 1e-8 relative over a 1 % change of lambda) — within 2 % AND no worse than the reference's own lambda by
    more than 4 fp32 ulps of the float64 objective."""
    if abs(lam - lam_ref) <= 1e-3 * lam_ref:
        return True
    f, f_ref = lbps_objective64(costs, lam, delta), lbps_objective64(costs, lam_ref, delta)
    return abs(lam - lam_ref) <= tol * lam_ref and f - f_ref <= 4 * float(np.finfo(np.float32).eps) * abs(f_ref)
