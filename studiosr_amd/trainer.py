"""Training loop with the reference `Trainer` surface (studiosr/engine/trainer.py:17-187) on the HIP training path.

What is kept, because callers and checkpoints depend on it: the constructor keywords (`model.get_training_config()` is splatted
into it, docs/README.md:30-34), `run()`, `evaluate()`, `build_optimizer()`, `save()` / `load()` with the reference's three files
(`<name>.model.pth` = state_dict, `<name>.train.pth` = optimizer / scheduler / iteration / best_psnr, `params.json` =
get_model_config()), Adam + MultiStepLR, the bf16 autocast context around forward + loss, DDP when RANK is set (one process per
GPU, RCCL all-reduce of the fp32 gradients, trainer.py:89-91) with the batch divided over ranks and seed + rank per process
(data/handler.py:42-72,86-88).  What differs: the forward / backward inside `step()` are HIP kernels (studiosr_amd/autograd.py);
the logger is plain print; the model never leaves the device it was built for.
"""
from __future__ import annotations

import json
import os
import random
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel
from torch.utils.data import DataLoader, Dataset, DistributedSampler

Tensor = torch.Tensor


class SyntheticPairs(Dataset):
    """Seeded uniform-random (LR, HR) patch pairs of the DIV2K training shape: no files, no network."""

    def __init__(self, scale: int = 4, lr_size: int = 64, length: int = 1 << 16, seed: int = 0) -> None:
        self.scale, self.lr_size, self.length, self.seed = scale, lr_size, length, seed

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, i: int) -> Tuple[Tensor, Tensor]:
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        s = self.lr_size
        return torch.rand(3, s, s, generator=g), torch.rand(3, s * self.scale, s * self.scale, generator=g)


class DataHandler:
    """Per-rank batches (data/handler.py:37-98): global batch / world size per process, DistributedSampler under DDP, endless iterator."""

    def __init__(self, dataset: Dataset, batch_size: int, num_workers: int, backend: Optional[str] = None) -> None:
        self.rank = int(os.environ.get("RANK", -1))
        self.local_rank = int(os.environ.get("LOCAL_RANK", 0))
        self.world_size = int(os.environ.get("WORLD_SIZE", 1))
        self.ddp_enabled = self.rank != -1
        self.owns_group = False
        if self.ddp_enabled:
            torch.cuda.set_device(self.local_rank)
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend or "nccl")  # RCCL on ROCm
                self.owns_group = True
            sampler = DistributedSampler(dataset, num_replicas=self.world_size, rank=self.rank, shuffle=True)
        else:
            self.rank, self.world_size, sampler = 0, 1, None
        self.is_main_process = self.rank == 0
        self.loader = DataLoader(dataset, batch_size=batch_size // self.world_size, num_workers=num_workers, sampler=sampler, shuffle=sampler is None,
                                 drop_last=True, pin_memory=True)
        self._it = iter(self.loader)
        self.iterations = 0

    def get_batch(self) -> Tuple[Tensor, Tensor]:
        try:
            batch = next(self._it)
        except StopIteration:
            self._it = iter(self.loader)
            batch = next(self._it)
        self.iterations += 1
        return batch

    def set_seed(self, seed: int) -> None:
        random.seed(seed + self.rank)
        torch.manual_seed(seed + self.rank)

    def set_iterations(self, iterations: int) -> None:
        self.iterations = iterations

    def close(self) -> None:
        if self.owns_group:
            dist.destroy_process_group()


class Trainer:
    def __init__(self, model: nn.Module, train_dataset: Dataset, evaluator=None, batch_size: int = 32, num_workers: int = 4, learning_rate: float = 0.0002,
                 beta1: float = 0.9, beta2: float = 0.99, weight_decay: float = 0.0, max_iters: int = 500000, gamma: float = 0.5,
                 milestones: List[int] = [250000, 400000, 450000, 475000], loss_function: Callable[[Tensor, Tensor], Tensor] = nn.L1Loss(),
                 eval_interval: int = 1000, ckpt_path: str = "checkpoints", bfloat16: bool = True, seed: int = 0, ddp_backend: Optional[str] = None) -> None:
        self.model, self.dataset, self.evaluator = model, train_dataset, evaluator
        self.batch_size, self.num_workers, self.max_iters, self.eval_interval = batch_size, num_workers, max_iters, eval_interval
        self.ckpt_path = ckpt_path
        os.makedirs(ckpt_path, exist_ok=True)
        self.learning_rate, self.betas, self.weight_decay, self.milestones, self.gamma = learning_rate, (beta1, beta2), weight_decay, milestones, gamma
        if not torch.cuda.is_available():
            raise RuntimeError("studiosr_amd.Trainer needs an MI355X: the HIP training path has no CPU fallback")
        self.dtype = torch.bfloat16 if bfloat16 else torch.float32
        self.seed, self.criterion, self.best_psnr = seed, loss_function, 0.0
        self.ddp_backend = ddp_backend
        self.optimizer = self.scheduler = self.data_handler = None
        self._wrapped: Optional[nn.Module] = None

    # ------------------------------------------------------------------ one optimisation step (trainer.py:97-109)
    def step(self, x: Tensor, y: Tensor) -> Tensor:
        """One optimisation step; returns the DETACHED loss tensor (no host sync: the reference loop reads the loss only inside its
        `eval_interval` branch, trainer.py:111-116, and a per-step `.item()` would serialise the host against ~1.4 k launches)."""
        model = self._wrapped if self._wrapped is not None else self.model
        with torch.autocast(device_type="cuda", dtype=self.dtype):
            out = model(x)
            loss = self.criterion(out, y)
        loss.backward()  # under DDP the RCCL all-reduce of the gradient buckets overlaps with the remaining backward kernels
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        self.scheduler.step()
        return loss.detach()

    def prepare(self) -> torch.device:
        """Everything run() does before its loop: data handler, seeds, device placement, checkpoint resume, DDP wrap."""
        self.data_handler = DataHandler(self.dataset, self.batch_size, self.num_workers, self.ddp_backend)
        self.data_handler.set_seed(self.seed)
        device = torch.device("cuda", self.data_handler.local_rank)
        self.model = self.model.to(device)
        if self.load("latest") and self.data_handler.is_main_process:
            print(f"-> The latest checkpoint was loaded. [best_psnr = {self.best_psnr:6.3f}]")
        self._wrapped = DistributedDataParallel(self.model, device_ids=[device.index], output_device=device.index) if self.data_handler.ddp_enabled else None
        self.model.train()
        return device

    def run(self) -> None:
        device = self.prepare()
        dh = self.data_handler
        log = open(os.path.join(self.ckpt_path, "train.log"), "a") if dh.is_main_process else None
        while dh.iterations < self.max_iters:
            x, y = dh.get_batch()
            loss = self.step(x.to(device, non_blocking=True), y.to(device, non_blocking=True))
            it = dh.iterations
            if it % self.eval_interval == 0 and dh.is_main_process:
                psnr, ssim = self.evaluate()
                line = f" Iterations = {it:<8}  loss: {float(loss):8.5f}  PSNR: {psnr:6.3f} SSIM: {ssim:6.4f}"
                print(line)
                log.write(line + "\n")
                log.flush()
                if self.best_psnr <= psnr:
                    self.best_psnr = psnr
                    self.save("best")
                self.save("latest")
        if log:
            log.close()
        dh.close()

    def evaluate(self) -> Tuple[float, float]:
        if not self.evaluator:
            return 0.0, 0.0
        self.model.eval()  # the UNWRAPPED module's inference(), outside autocast: the reference-precision HIP path (trainer.py:125-131)
        psnr, ssim = self.evaluator.run(self.model.inference)
        self.model.train()
        return psnr, ssim

    def build_optimizer(self):
        from .optim import Adam  # torch.optim.Adam; one flat launch when the fused training path (fasttrain.py) owns the parameters

        opt = Adam(self.model.parameters(), model=self.model, lr=self.learning_rate, betas=self.betas, weight_decay=self.weight_decay)
        return opt, torch.optim.lr_scheduler.MultiStepLR(opt, milestones=self.milestones, gamma=self.gamma)

    # ------------------------------------------------------------------ checkpoints (trainer.py:148-187)
    def _paths(self, name: str) -> Tuple[str, str]:
        return os.path.join(self.ckpt_path, name + ".model.pth"), os.path.join(self.ckpt_path, name + ".train.pth")

    def save(self, file_name: str) -> Tuple[str, str]:
        os.makedirs(self.ckpt_path, exist_ok=True)
        model_path, train_path = self._paths(file_name)
        torch.save(self.model.state_dict(), model_path)
        torch.save(dict(optimizer=self.optimizer.state_dict(), scheduler=self.scheduler.state_dict(),
                        iteration=self.data_handler.iterations if self.data_handler else 0, best_psnr=self.best_psnr), train_path)
        with open(os.path.join(self.ckpt_path, "params.json"), "w") as f:
            json.dump(self.model.get_model_config(), f)
        return model_path, train_path

    def load(self, file_name: str) -> bool:
        model_path, train_path = self._paths(file_name)
        self.optimizer, self.scheduler = self.build_optimizer()
        if not (os.path.isfile(model_path) and os.path.isfile(train_path)):
            return False
        device = next(self.model.parameters()).device
        state: Dict = torch.load(train_path, map_location=device)
        self.model.load_state_dict(torch.load(model_path, map_location=device))
        self.optimizer.load_state_dict(state["optimizer"])
        self.scheduler.load_state_dict(state["scheduler"])
        if self.data_handler:
            self.data_handler.set_iterations(state["iteration"])
        self.best_psnr = state.get("best_psnr", 0.0)
        return True
