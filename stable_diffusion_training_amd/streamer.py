"""Synthetic stand-in for the reference's `streamer.dataloader.DataLoader` (an un-vendored submodule: `streamer/` is empty
under /root/reference; its surface is what training.py:49-81, 122-139, 193-203 uses).  Same constructor arguments and the
same methods / attributes the loop touches, but the images and captions are seeded noise of the shapes the real loader
emits: aspect-ratio buckets from `calculate_resolution_array` (training_utils.py:134-174), `repeat_batch` consecutive
batches per bucket (so the step dispatcher does not bounce between shapes), k concatenated 77-token caption windows.

Data-parallel use: every rank builds the loader with the same seed and its (rank, world_size); all ranks walk the same
bucket sequence and each materialises its own shard of the GLOBAL batch (training_utils.py:805).  SURVEY.md §8(f)4.
"""
import numpy as np
import torch

from .training_utils import calculate_resolution_array


class DataLoader:
    def __init__(self, tokenizer_obj=None, config=None, ramdisk_path=None, training_batch_size=8, repeat_batch=10,
                 maximum_resolution_areas=(512 ** 2,), bucket_lower_bound_resolutions=(256,), numb_of_worker_thread=1,
                 queue_get_timeout=60, chunk_number=0, seed=0, context_concatenation_multiplier=1, *,
                 batches_per_chunk=100, vocab_size=49408, context_window=77, rank=0, world_size=1, device="cpu"):
        if len(maximum_resolution_areas) != len(bucket_lower_bound_resolutions):
            raise ValueError("number of elements in maximum_resolution_areas and bucket_lower_bound_resolutions is not match!")
        if training_batch_size % world_size:
            raise ValueError(f"global batch {training_batch_size} is not divisible by {world_size} ranks")
        self.tokenizer_obj, self.config, self.ramdisk_path = tokenizer_obj, config, ramdisk_path
        self.training_batch_size, self.repeat_batch = training_batch_size, max(int(repeat_batch), 1)
        self.chunk_number, self.seed = chunk_number, seed
        self.k, self.vocab_size, self.context_window = context_concatenation_multiplier, vocab_size, context_window
        self.rank, self.world_size, self.device = rank, world_size, torch.device(device)
        self.buckets = [tuple(int(v) for v in b) for area, lo in zip(maximum_resolution_areas, bucket_lower_bound_resolutions)
                        for b in calculate_resolution_array(area, lo, 64)]
        self._batches_per_chunk = batches_per_chunk
        self._bulk_batch_count, self._first_batch_count = 0, 0
        self._print_debug = True
        self._plan, self._cursor = [], 0

    # ---- chunk bookkeeping the loop calls (no files behind it here)
    def delete_prev_chunks(self, prev_chunk):
        return None

    def grab_and_prefetch_chunk(self, numb_of_prefetched_batch=1):
        return None

    def prepare_training_dataframe(self):
        return None

    def create_training_dataframe(self):
        """Lay out this chunk's batches: runs of `repeat_batch` batches per bucket, bucket order shuffled by (seed, chunk)."""
        rng = np.random.default_rng([int(self.seed), int(self.chunk_number)])
        plan = []
        while len(plan) < self._batches_per_chunk:
            plan.extend([self.buckets[int(rng.integers(len(self.buckets)))]] * self.repeat_batch)
        self._plan = plan[: self._batches_per_chunk]
        self._first_batch_count, self._bulk_batch_count = min(self.repeat_batch, len(self._plan)), max(len(self._plan) - self.repeat_batch, 0)

    def dispatch_worker(self):
        self._cursor = 0

    # ---- batches
    def grab_next_batch(self):
        """dict(pixel_values f32 (B,3,bucket[0],bucket[1]) in [-1,1], input_ids / attention_mask int32 (B, k*77)) for this
        rank's shard, or "end_of_batch" when the chunk is exhausted (training.py:195-199)."""
        if self._cursor >= len(self._plan):
            return "end_of_batch"
        b0, b1 = self._plan[self._cursor]
        index = self._cursor
        self._cursor += 1
        per_rank = self.training_batch_size // self.world_size
        g = torch.Generator().manual_seed((int(self.seed) * 1_000_003 + int(self.chunk_number)) * 1_000_003 + index * 64 + self.rank)
        px = torch.rand(per_rank, 3, b0, b1, generator=g) * 2 - 1
        ids = torch.randint(0, self.vocab_size - 2, (per_rank, self.k, self.context_window), generator=g, dtype=torch.int32)
        ids[:, :, 0] = self.vocab_size - 2   # <|startoftext|>
        ids[:, :, -1] = self.vocab_size - 1  # <|endoftext|>
        ids = ids.reshape(per_rank, self.k * self.context_window)
        batch = {"pixel_values": px, "input_ids": ids, "attention_mask": torch.ones_like(ids)}
        if self.device.type != "cpu":
            batch = {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}
        if self._print_debug:
            print(f"[streamer] chunk {self.chunk_number} batch {index}: {tuple(px.shape)}")
        return batch
