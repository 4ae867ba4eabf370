"""A builder-written transcription of what a user of the reference's pendulum example holds (example/pendulum.py:10-47): a
TorchScript `dynamics` nested in a function and a plain-Python cost closure over a module-level TorchScript angle wrap — the
gymnasium Pendulum-v1 step and its quadratic cost, written here from the model's equations with this
repo's own names and layout.  It carries NO native tag: pi_mpc/recognize.py must recognise it through the version-independent
("g2") fingerprints of the shipped table — the operator sequence of the scripted graph and the structure of the cost's AST —
on whatever torch the machine runs, and through its behaviour on the probe batches.  Test input only."""
import torch


@torch.jit.script
def wrap_to_pi(angle):
    return ((angle + torch.pi) % (2 * torch.pi)) - torch.pi


def build():
    @torch.jit.script
    def step(x: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
        ang = x[:, 0].view(-1, 1)
        rate = x[:, 1].view(-1, 1)
        g = 10
        m = 1
        arm = 1
        h = 0.05
        torque = u[:, 0].view(-1, 1)
        torque = torch.clamp(torque, -2, 2)
        rate_next = rate + (-3 * g / (2 * arm) * torch.sin(ang + torch.pi) + 3.0 / (m * arm**2) * torque) * h
        ang_next = ang + rate_next * h
        rate_next = torch.clamp(rate_next, -8, 8)
        x = torch.cat((ang_next, rate_next), dim=1)
        return x

    def running_cost(x: torch.Tensor, u: torch.Tensor, extra) -> torch.Tensor:
        ang = x[:, 0]
        rate = x[:, 1]
        c = wrap_to_pi(ang) ** 2 + 0.1 * rate**2
        return c

    return step, running_cost
