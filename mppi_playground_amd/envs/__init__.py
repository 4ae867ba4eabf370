"""The shipped model plugins (dynamics / cost callables) and their host-side set-up code.

Each callable keeps the reference's contract — dynamics(state[B,ds], action[B,dc]) -> [B,ds],
cost(state[B,ds], action[B,dc], info) -> [B] in torch — and additionally carries a native tag
(pi_mpc/native.py) so that pi_mpc.mppi.MPPI runs it fused on the device.  Rendering / video /
gymnasium simulators of the reference are UI and out of scope; `render()`/`close()` are no-ops.
"""
