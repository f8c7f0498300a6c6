"""Console / wandb reporting of one iteration (role of rlgym_ppo/util/reporting.py; outside the accelerated path).
The grouping below is the reference's report contract: the keys PPOLearner.learn and Learner._learn must emit."""
import numpy as np
import torch

GROUPS = (
    ("Policy Reward", "Policy Entropy", "Value Function Loss"),
    ("Mean KL Divergence", "SB3 Clip Fraction", "Policy Update Magnitude", "Value Function Update Magnitude"),
    ("Collected Steps per Second", "Overall Steps per Second"),
    ("Timestep Collection Time", "Timestep Consumption Time", "PPO Batch Consumption Time", "Total Iteration Time"),
    ("Cumulative Model Updates", "Cumulative Timesteps"),
    ("Timesteps Collected",),
)


def _form_printable_groups(report):
    return [{k: report[k] for k in keys} for keys in GROUPS]  # KeyError if the hot path forgot a key


def _fmt(val):
    if isinstance(val, torch.Tensor):
        val = val.detach().cpu().item() if val.dim() == 0 else val.detach().cpu().tolist()
    if isinstance(val, (tuple, list, np.ndarray)):
        return "[" + ", ".join(f"{v:7.5f}" if isinstance(v, float) else str(v) for v in val) + "]"
    if isinstance(val, (float, np.floating)):
        return f"{val:,.5f}"
    if isinstance(val, (int, np.integer)):
        return f"{val:,d}"
    return str(val)


def dump_dict_to_debug_string(dictionary):
    return "".join(f"{k}: {_fmt(v)}\n" for k, v in dictionary.items())


def report_metrics(loggable_metrics, debug_metrics, wandb_run=None):
    if wandb_run is not None:
        wandb_run.log(loggable_metrics)
    if debug_metrics is not None:
        print("\nBEGIN DEBUG\n")
        print(dump_dict_to_debug_string(debug_metrics))
        print("\nEND DEBUG\n")
    print("{}{}{}".format("-" * 8, "BEGIN ITERATION REPORT", "-" * 8))
    out = "".join(dump_dict_to_debug_string(g) + "\n" for g in _form_printable_groups(loggable_metrics))
    print(out[:-2])
    print("{}{}{}\n\n".format("-" * 8, "END ITERATION REPORT", "-" * 8))
