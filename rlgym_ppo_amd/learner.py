"""Learner -- drop-in for rlgym_ppo.Learner (reference: rlgym_ppo/learner.py:28-576).

Same 40 constructor parameters, same loop (collect -> add_new_experience -> PPOLearner.learn -> report -> checkpoint),
same checkpoint folder layout and BOOK_KEEPING_VARS.json.  The three hot calls of the loop body (learner.py:257-270)
run on the GPU through librlppo.so: rollout inference inside BatchedAgentManager._send_actions, value pass + GAE in
add_new_experience (device-resident, no tolist()), and the PPO update.
"""
import json
import os
import random
import shutil
import time

import numpy as np
import torch

from .batched_agents import BatchedAgentManager, VectorAgentManager
from .ppo import ExperienceBuffer, PPOLearner
from .util import KBHit, WelfordRunningStat, reporting, torch_functions

try:  # wandb is optional (not installed on the build/GPU boxes)
    import wandb
except ImportError:  # pragma: no cover
    wandb = None


class Learner(object):
    def __init__(
            self, env_create_function, metrics_logger=None, n_proc: int = 8, min_inference_size: int = 80,
            render: bool = False, render_delay: float = 0,
            timestep_limit: int = 5_000_000_000, exp_buffer_size: int = 100000, ts_per_iteration: int = 50000,
            standardize_returns: bool = True, standardize_obs: bool = True, max_returns_per_stats_increment: int = 150,
            steps_per_obs_stats_increment: int = 5,
            policy_layer_sizes=(256, 256, 256), critic_layer_sizes=(256, 256, 256), continuous_var_range=(0.1, 1.0),
            ppo_epochs: int = 10, ppo_batch_size: int = 50000, ppo_minibatch_size=None, ppo_ent_coef: float = 0.005,
            ppo_clip_range: float = 0.2, gae_lambda: float = 0.95, gae_gamma: float = 0.99, policy_lr: float = 3e-4,
            critic_lr: float = 3e-4,
            log_to_wandb: bool = False, load_wandb: bool = True, wandb_run=None, wandb_project_name=None,
            wandb_group_name=None, wandb_run_name=None,
            checkpoints_save_folder=None, add_unix_timestamp: bool = True, checkpoint_load_folder="latest",
            save_every_ts: int = 1_000_000, instance_launch_delay=None, random_seed: int = 123,
            n_checkpoints_to_keep: int = 5, shm_buffer_size: int = 8192, device: str = "auto",
            vector_env: bool = False, per_feature_obs_standardization: bool = False):
        assert env_create_function is not None, "MUST PROVIDE A FUNCTION TO CREATE RLGYM FUNCTIONS TO INITIALIZE RLGYM-PPO"
        if checkpoints_save_folder is None:
            checkpoints_save_folder = os.path.join("data", "checkpoints", "rlgym-ppo-run")
        self.add_unix_timestamp = add_unix_timestamp
        if add_unix_timestamp:
            checkpoints_save_folder = f"{checkpoints_save_folder}-{time.time_ns()}"

        torch.manual_seed(random_seed)
        np.random.seed(random_seed)
        random.seed(random_seed)

        self.n_checkpoints_to_keep = n_checkpoints_to_keep
        self.checkpoints_save_folder = checkpoints_save_folder
        self.max_returns_per_stats_increment = max_returns_per_stats_increment
        self.metrics_logger = metrics_logger
        self.standardize_returns = standardize_returns
        self.save_every_ts = save_every_ts
        self.ts_since_last_save = 0

        if device in {"auto", "gpu"}:
            if not torch.cuda.is_available():
                raise RuntimeError("rlgym_ppo_amd.Learner needs an AMD GPU visible to PyTorch (no CPU fallback)")
            self.device = f"cuda:{int(os.environ.get('LOCAL_RANK', 0))}"
        else:
            self.device = device
        torch.cuda.set_device(torch.device(self.device))
        print(f"Using device {self.device}")

        self.exp_buffer_size = exp_buffer_size
        self.timestep_limit = timestep_limit
        self.ts_per_epoch = ts_per_iteration
        self.gae_lambda = gae_lambda
        self.gae_gamma = gae_gamma
        self.return_stats = WelfordRunningStat(1)
        self.epoch = 0
        # construction order buffer -> manager -> processes -> PPOLearner fixes RNG consumption (learner.py:124-167)
        self.experience_buffer = ExperienceBuffer(self.exp_buffer_size, seed=random_seed, device=self.device)

        print("Initializing processes...")
        collect_metrics_fn = None if metrics_logger is None else self.metrics_logger.collect_metrics
        # vector_env=True (not in the reference): env_create_function builds ONE vectorised environment whose agents step in
        # lockstep; the rollout then stays on the GPU (batched_agents/vector_agent_manager.py)
        manager_cls = VectorAgentManager if vector_env else BatchedAgentManager
        self.agent = manager_cls(None, min_inference_size=min_inference_size, seed=random_seed,
                                 standardize_obs=standardize_obs,
                                 steps_per_obs_stats_increment=steps_per_obs_stats_increment)
        # not in the reference (which standardises every feature with the statistics of feature 0, quirk Q5): every feature
        # with its own running mean / std
        self.agent.per_feature_obs_standardization = bool(per_feature_obs_standardization)
        obs_space_size, act_space_size, action_space_type = self.agent.init_processes(
            n_processes=n_proc, build_env_fn=env_create_function, collect_metrics_fn=collect_metrics_fn,
            spawn_delay=instance_launch_delay, render=render, render_delay=render_delay, shm_buffer_size=shm_buffer_size)
        obs_space_size = np.prod(obs_space_size)
        print("Initializing PPO...")
        if ppo_minibatch_size is None:
            ppo_minibatch_size = ppo_batch_size
        self.ppo_learner = PPOLearner(
            obs_space_size, act_space_size, device=self.device, batch_size=ppo_batch_size,
            mini_batch_size=ppo_minibatch_size, n_epochs=ppo_epochs, continuous_var_range=continuous_var_range,
            policy_type=action_space_type, policy_layer_sizes=policy_layer_sizes, critic_layer_sizes=critic_layer_sizes,
            policy_lr=policy_lr, critic_lr=critic_lr, clip_range=ppo_clip_range, ent_coef=ppo_ent_coef)
        self.agent.policy = self.ppo_learner.policy

        self.config = {
            "n_proc": n_proc, "min_inference_size": min_inference_size, "timestep_limit": timestep_limit,
            "exp_buffer_size": exp_buffer_size, "ts_per_iteration": ts_per_iteration,
            "standardize_returns": standardize_returns, "standardize_obs": standardize_obs,
            "policy_layer_sizes": policy_layer_sizes, "critic_layer_sizes": critic_layer_sizes, "ppo_epochs": ppo_epochs,
            "ppo_batch_size": ppo_batch_size, "ppo_minibatch_size": ppo_minibatch_size, "ppo_ent_coef": ppo_ent_coef,
            "ppo_clip_range": ppo_clip_range, "gae_lambda": gae_lambda, "gae_gamma": gae_gamma, "policy_lr": policy_lr,
            "critic_lr": critic_lr, "shm_buffer_size": shm_buffer_size,
        }

        self.wandb_run = wandb_run
        wandb_loaded = checkpoint_load_folder is not None and self.load(checkpoint_load_folder, load_wandb, policy_lr, critic_lr)
        if log_to_wandb and self.wandb_run is None and not wandb_loaded:
            if wandb is None:
                raise RuntimeError("log_to_wandb=True but the wandb package is not installed")
            self.wandb_run = wandb.init(project=wandb_project_name or "rlgym-ppo", group=wandb_group_name or "unnamed-runs",
                                        config=self.config, name=wandb_run_name or "rlgym-ppo-run", reinit=True)
            print("Created new wandb run!", self.wandb_run.id)
        print("Learner successfully initialized!")

    def update_learning_rate(self, new_policy_lr=None, new_critic_lr=None):
        if new_policy_lr is not None:
            self.policy_lr = new_policy_lr
            for group in self.ppo_learner.policy_optimizer.param_groups:
                group["lr"] = new_policy_lr
            print(f"New policy learning rate: {new_policy_lr}")
        if new_critic_lr is not None:
            self.critic_lr = new_critic_lr
            for group in self.ppo_learner.value_optimizer.param_groups:
                group["lr"] = new_critic_lr
            print(f"New critic learning rate: {new_critic_lr}")

    def learn(self):
        """Run the loop; on any error print it, try to checkpoint, always clean up (learner.py:218-238)."""
        try:
            self._learn()
        except Exception:
            import traceback
            print("\n\nLEARNING LOOP ENCOUNTERED AN ERROR\n")
            traceback.print_exc()
            try:
                self.save(self.agent.cumulative_timesteps)
            except Exception:
                print("FAILED TO SAVE ON EXIT")
        finally:
            self.cleanup()

    def _learn(self):
        kb = KBHit()
        print("Press (p) to pause (c) to checkpoint, (q) to checkpoint and quit (after next iteration)\n")
        while self.agent.cumulative_timesteps < self.timestep_limit:
            epoch_start = time.perf_counter()
            experience, collected_metrics, steps_collected, collection_time = self.agent.collect_timesteps(self.ts_per_epoch)
            if self.metrics_logger is not None:
                self.metrics_logger.report_metrics(collected_metrics, self.wandb_run, self.agent.cumulative_timesteps)
            self.add_new_experience(experience)
            report = dict(self.ppo_learner.learn(self.experience_buffer))
            epoch_time = time.perf_counter() - epoch_start
            if self.epoch < 1:
                report["Value Function Loss"] = np.nan  # quirk Q10
            report["Cumulative Timesteps"] = self.agent.cumulative_timesteps
            report["Total Iteration Time"] = epoch_time
            report["Timesteps Collected"] = steps_collected
            report["Timestep Collection Time"] = collection_time
            report["Timestep Consumption Time"] = epoch_time - collection_time
            report["Collected Steps per Second"] = steps_collected / collection_time
            report["Overall Steps per Second"] = steps_collected / epoch_time
            report["Policy Reward"] = self.agent.average_reward if self.agent.average_reward is not None else np.nan
            self.ts_since_last_save += steps_collected
            reporting.report_metrics(loggable_metrics=report, debug_metrics=None, wandb_run=self.wandb_run)

            if kb.kbhit():
                c = kb.getch()
                if c == "p":
                    print("Paused, press any key to resume")
                    while not kb.kbhit():
                        time.sleep(0.05)
                if c in ("c", "q"):
                    self.save(self.agent.cumulative_timesteps)
                if c == "q":
                    return
                if c in ("c", "p"):
                    print("Resuming...\n")
            if self.ts_since_last_save >= self.save_every_ts:
                self.save(self.agent.cumulative_timesteps)
                self.ts_since_last_save = 0
            self.epoch += 1

    @torch.no_grad()
    def add_new_experience(self, experience):
        """Value pass on [N+1, d] (states + the last next_state, learner.py:347-349), GAE, return statistics, buffer
        submit -- all on the device; only min(150, N) returns come back to the host (learner.py:368-372)."""
        states, actions, log_probs, rewards, next_states, dones, truncated = experience
        value_net = self.ppo_learner.value_net
        n = states.shape[0]
        on_device = isinstance(states, torch.Tensor) and states.is_cuda   # VectorAgentManager: rollout already in HBM
        if on_device:
            rows = self.agent.value_input_rows               # [N+1, ld]: states ++ the last next_state, padded
            assert rows.shape[0] == n + 1 and rows.data_ptr() == states.data_ptr()
            d_logical = int(self.ppo_learner.policy.arena.d_in)
        else:
            val_inp = np.concatenate([np.asarray(states).reshape(n, -1), np.asarray(next_states[-1]).reshape(1, -1)], axis=0)
            rows = value_net.arena.stage_obs(val_inp)        # zero-padded fp32 device rows [N+1, ld]
            d_logical = int(np.asarray(states).reshape(n, -1).shape[1])
        val_preds = value_net.forward_padded(rows).contiguous()

        dev = rows.device
        up = lambda x: x.to(dev, torch.float32) if isinstance(x, torch.Tensor) else \
            torch.as_tensor(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(dev)
        rews_d, dones_d, trunc_d = up(rewards), up(dones), up(truncated)
        ret_std = self.return_stats.std[0] if self.standardize_returns else None
        value_targets, advantages, returns, gae_timeouts = torch_functions.gae_device_deferred(
            rews_d, dones_d, trunc_d, val_preds, gamma=self.gae_gamma, lmbda=self.gae_lambda, return_std=ret_std)

        # ONE device-to-host read per call: the scan's timeout counter travels with the min(150, N) returns the statistics need
        n_to_increment = min(self.max_returns_per_stats_increment, n) if self.standardize_returns else 0
        parts = [returns[:n_to_increment]] + ([gae_timeouts.float()] if gae_timeouts is not None else [])
        host = torch.cat(parts).cpu().numpy()
        if gae_timeouts is not None:
            torch_functions.raise_if_timed_out(host[-1])   # GAETimeout: NaN advantages must never reach the buffer
        if self.standardize_returns:
            head = host[:n_to_increment]
            if np.isnan(head).any():
                raise RuntimeError("GAE produced NaN returns (NaN inputs): refusing to train on them")
            self.return_stats.increment(head, n_to_increment)

        self.experience_buffer._d = d_logical  # logical width of the padded rows
        self.experience_buffer.submit_experience(rows[:n], actions, log_probs, rews_d, next_states, dones_d, trunc_d,
                                                 value_targets, advantages)

    # ------------------------------------------------------------------------------------------ checkpoints
    def save(self, cumulative_timesteps):
        from .dp import dist_info
        if dist_info()[1] != 0:
            return  # data-parallel replicas are bit-identical: rank 0 writes the checkpoint
        folder_path = os.path.join(self.checkpoints_save_folder, str(cumulative_timesteps))
        os.makedirs(folder_path, exist_ok=True)
        print(f"Saving checkpoint {cumulative_timesteps}...")
        existing = sorted(int(name) for name in os.listdir(self.checkpoints_save_folder) if name.isdigit())
        if len(existing) > self.n_checkpoints_to_keep:
            for name in existing[:-self.n_checkpoints_to_keep]:
                shutil.rmtree(os.path.join(self.checkpoints_save_folder, str(name)))
        os.makedirs(folder_path, exist_ok=True)
        self.ppo_learner.save_to(folder_path)
        book = {
            "cumulative_timesteps": self.agent.cumulative_timesteps,
            "cumulative_model_updates": self.ppo_learner.cumulative_model_updates,
            "policy_average_reward": self.agent.average_reward,
            "epoch": self.epoch,
            "ts_since_last_save": self.ts_since_last_save,
            "reward_running_stats": self.return_stats.to_json(),
        }
        if self.agent.standardize_obs and self.agent.obs_stats is not None:
            book["obs_running_stats"] = self.agent.obs_stats.to_json()
        if self.wandb_run is not None:
            book.update(wandb_run_id=self.wandb_run.id, wandb_project=self.wandb_run.project,
                        wandb_entity=self.wandb_run.entity, wandb_group=self.wandb_run.group,
                        wandb_config=self.wandb_run.config.as_dict())
        with open(os.path.join(folder_path, "BOOK_KEEPING_VARS.json"), "w") as f:
            json.dump(book, f, indent=4)
        print(f"Checkpoint {cumulative_timesteps} saved!\n")

    def _latest_checkpoint(self):
        base = self.checkpoints_save_folder
        if base is None:
            return None
        if self.add_unix_timestamp:
            stem = base[:base.rfind("-")]
            parent = os.path.dirname(stem)
            if not os.path.exists(parent):
                return None
            runs = []
            for name in os.listdir(parent):
                full = os.path.join(parent, name)
                stamp = name[name.rfind("-") + 1:]
                if os.path.isdir(full) and full.startswith(stem) and "-" in name and stamp.isdigit():
                    runs.append((int(stamp), full))
            if not runs:
                return None
            base = max(runs)[1]
        elif not os.path.exists(base):
            return None
        steps = [int(name) for name in os.listdir(base) if name.isdigit() and os.path.isdir(os.path.join(base, name))]
        return os.path.join(base, str(max(steps))) if steps else None

    def load(self, folder_path, load_wandb, new_policy_lr=None, new_critic_lr=None):
        if folder_path == "latest":
            folder_path = self._latest_checkpoint()
            if folder_path is None:
                return
            print(f"Auto-load path: {folder_path}")
        assert os.path.exists(folder_path), f"UNABLE TO LOCATE FOLDER {folder_path}"
        print(f"Loading from checkpoint at {folder_path}")
        self.ppo_learner.load_from(folder_path)
        wandb_loaded = False
        with open(os.path.join(folder_path, "BOOK_KEEPING_VARS.json"), "r") as f:
            book = dict(json.load(f))
        self.agent.cumulative_timesteps = book["cumulative_timesteps"]
        self.agent.average_reward = book["policy_average_reward"]
        self.ppo_learner.cumulative_model_updates = book["cumulative_model_updates"]
        self.return_stats.from_json(book["reward_running_stats"])
        if self.agent.standardize_obs and "obs_running_stats" in book:
            self.agent.obs_stats = WelfordRunningStat(1)
            self.agent.obs_stats.from_json(book["obs_running_stats"])
        self.epoch = book["epoch"]
        if new_policy_lr is not None or new_critic_lr is not None:
            self.update_learning_rate(new_policy_lr, new_critic_lr)
        if "wandb_run_id" in book and load_wandb and wandb is not None:
            self.wandb_run = wandb.init(settings=wandb.Settings(start_method="spawn"), entity=book["wandb_entity"],
                                        project=book["wandb_project"], group=book["wandb_group"], id=book["wandb_run_id"],
                                        config=book["wandb_config"], resume="allow", reinit=True)
            wandb_loaded = True
        print("Checkpoint loaded!")
        return wandb_loaded

    def cleanup(self):
        if self.wandb_run is not None:
            self.wandb_run.finish()
        if type(self.agent) in (BatchedAgentManager, VectorAgentManager):
            self.agent.cleanup()
        self.experience_buffer.clear()
