"""The training / evaluation hot loop around the model (`corenet.pipeline`, pipeline.py:100-242,311-325):
data loader, ground-truth voxelization settings, and one training step from a list of dataset elements.
Configuration objects, TensorBoard, checkpoint management and the progress UI of the reference pipeline
are out of scope; their values arrive here as plain arguments."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch as t

from corenet_amd import distributed as dist_util
from corenet_amd.data import batched_example, dataset as dataset_lib

LOSS_OF_TASK = {"fg_bg": "iou_fgbg", "semantic": "xent_times_iou_agnostic"}      # pipeline.py:150-154


def create_distributed_loader(dataset, batch_size: int, global_rank: int, global_world_size: int,
                              num_data_workers: int = 0, prefetch_factor: Optional[int] = None,
                              pad_data: bool = False) -> t.utils.data.DataLoader:
  """pipeline.py:100-119: the loader hands over *lists* of elements (`collate_fn=lambda v: v`); `batch`
  runs in the training process, which is where this build sends the geometry to the GPU."""
  sampler = dist_util.DistributedSampler(dataset, global_rank=global_rank, global_world_size=global_world_size,
                                         pad_data=pad_data)
  ctx = t.multiprocessing.get_context("fork") if num_data_workers > 0 else None
  return t.utils.data.DataLoader(
      dataset, batch_size=batch_size, num_workers=num_data_workers, pin_memory=True, collate_fn=_identity,
      sampler=sampler, multiprocessing_context=ctx, drop_last=False,
      prefetch_factor=prefetch_factor if num_data_workers > 0 else None)


def _identity(v):
  return v


def voxelize_batch(b: batched_example.BatchedExample, task_type: str = "fg_bg",
                   resolution: Tuple[int, int, int] = (128, 128, 128), sub_grid_sampling: bool = False,
                   voxelization_image_resolution_multiplier=8, conservative_rasterization: bool = False,
                   voxelization_projection_depth_multiplier: int = 1) -> batched_example.BatchedExample:
  """pipeline.py:122-142 with the fields of `VoxelizationConfig` as arguments (defaults: configs/models/h7.json5)."""
  voxel_content_fn = {"semantic": batched_example.VoxelContentSemanticLabel(b.mesh_labels),
                      "fg_bg": batched_example.voxel_content_1}[task_type]
  return batched_example.voxelize(
      b, resolution=tuple(resolution), voxel_content_fn=voxel_content_fn, sub_grid_sampling=sub_grid_sampling,
      image_resolution_multiplier=voxelization_image_resolution_multiplier,
      conservative_rasterization=conservative_rasterization,
      projection_depth_multiplier=voxelization_projection_depth_multiplier)


def process_batch(model, batch: List[dataset_lib.DatasetElement], task_type: str = "fg_bg", lr: float = 4e-4,
                  adam_eps: float = 1e-4, world_size: int = 1, all_reduce=None, **voxelization) -> t.Tensor:
  """One training step from dataset elements (TrainPipeline._process_batch, pipeline.py:215-242): batch ->
  ground-truth grid -> `v2s = camera @ inverse(v2x)` -> forward, loss, backward, gradient exchange, Adam
  (`CoreNet.train_step`).  Returns the loss as a device tensor (the reference syncs on it every step)."""
  if world_size > 1 and getattr(all_reduce, "engine", None) is not model.engine:
    # DistributedDataParallel(broadcast_buffers=True) semantics (pipeline.py:199): every forward starts from rank
    # 0's BatchRenorm running statistics, which feed the r / d clamps (batch_renorm.py:46-49).  An exchange that was
    # attached to this model (`GradientSync(world).attach(model.engine)`) delivers exactly that on the first gradient
    # bucket of the previous step, without a collective of its own: the blocking broadcast in front of the step is only
    # for exchange objects that do not
    dist_util.broadcast_buffers(model.engine.store)
  ex = batched_example.batch(batch, device=model.engine.device)
  ex = voxelize_batch(ex, task_type, **voxelization)
  # v2s = camera @ inverse(v2x) (pipeline.py:219-221).  v2x is scale(m, m, m) built from the host-side resolution
  # (batched_example.voxelize), so its inverse is taken on the host copy of the same 4x4 and uploaded: no
  # device -> host read-back inside the step
  m = float(max(voxelization.get("resolution", (128, 128, 128))))
  inv_v2x = t.diag(t.tensor([m, m, m, 1.0])).inverse().to(ex.camera_transform.device, non_blocking=True)
  v2s = ex.camera_transform @ inv_v2x
  return model.train_step(ex.input_image, v2s, ex.grid_sampling_offset, ex.grid, LOSS_OF_TASK[task_type],
                          lr=lr, adam_eps=adam_eps, world_size=world_size, all_reduce=all_reduce)


def evaluate_batch(inference_fn, batch: List[dataset_lib.DatasetElement], task_type: str = "fg_bg", device="cuda",
                   **voxelization):
  """The body of EvalPipeline.run_eval's loop (pipeline.py:311-320): returns (pmf, voxelized batch)."""
  ex = batched_example.batch(batch, device=device)
  ex = voxelize_batch(ex, task_type, **voxelization)
  resolution = tuple(ex.grid.shape[1:])
  with t.no_grad():
    pmf = inference_fn(ex.input_image, ex.camera_transform, ex.v2x_transform, ex.grid_sampling_offset, resolution)
  return pmf, ex
