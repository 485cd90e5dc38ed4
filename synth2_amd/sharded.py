"""Voice-pool sharding over the GPUs of one node: one process per GPU, torch.distributed for the
single small exchange per buffer.

Voices never interact before the mix (the reference adds each voice's 16-frame buffer into an
accumulator, synth.rs:177-195), so the pool is dealt out to the ranks in runs of 64 voices
(round-robin; contiguous ranges with ``interleave=0``).  Every
rank sees the SAME note-event stream and runs the same voice-allocation policy over the whole
pool (host-side, cheap); only events that land in its own range reach its GPU.  Per buffer each
rank renders a partial mix (frames x 4 B = 4 KiB at 1024 frames), the partials are all-gathered
(RCCL over xGMI; latency-bound, not bandwidth-bound) and rank 0 adds them in RANK ORDER, rooted
at +0.0 (with contiguous ranges that is the association a single GPU produces with
``mix_groups = world``).

``reduce_to_root=True`` swaps the all-gather + ordered sum for ONE ``reduce(SUM)`` to rank 0 — the portable
fallback SURVEY §8(e) names.  The library then chooses the association, so the result is the rank-ordered sum bit for
bit only where the order cannot matter (two ranks: a + b == b + a); with more ranks it agrees to rounding, not to the
bit, which is why the all-gather stays the default (tests/test_sharded_gloo.py shows both facts).

The renderer and the row-combine are injectable so the exchange logic can be exercised on CPU
with gloo (tests/test_sharded_gloo.py feeds it partial rows made by the CPU oracle); the
defaults are the HIP path and nothing else — there is no CPU fallback here.

``devices=[d0, d1, ...]`` (one process, several GPUs) hands the whole job to ONE libs2r handle over a device list
(s2r_config.devices: the allocation policy run once per event, every shard launched on its own device's stream, the rows
written peer-to-peer into d0's buffer and added there in shard order) — no torch.distributed at all; this class is then a
thin wrapper around it.

Of the RCCL path what a one-GPU box can run has run: ONE rank through ``backend="nccl"`` (``force_exchange``:
tests/test_gpu_parity.py::test_sharded_exchange_through_rccl_with_one_rank — all-gather with and without overlap, and the
reduce, bit-identical to the plain path) and N ranks under gloo; the first N-GPU execution is the driver's bench.
"""
import torch
import torch.distributed as dist

from . import synth as _synth


class ShardedSynth:
    def __init__(self, voices_per_rank, max_frames=1024, rank=0, world=1, device=None, renderer=None,
                 combine=None, block_voices=0, overlap=True, interleave=64, reduce_to_root=False, force_exchange=False, devices=None):
        self.devices = list(devices) if devices else None
        if self.devices:
            # one process over a device list: everything happens inside the C ABI (s2r_config.devices)
            assert world == 1 and renderer is None
            n = len(self.devices)
            self.rank, self.world = 0, 1
            self.voices_per_rank = voices_per_rank
            self.total_voices = voices_per_rank * n
            self.max_frames = max_frames
            self.device = torch.device("cuda", self.devices[0])
            self.renderer = _synth.Synth(self.total_voices, max_frames=max_frames, block_voices=block_voices,
                                         shard_interleave=interleave if n > 1 else 0, devices=self.devices)
            self._host_target = None
            self._out = None
            return
        self.rank, self.world = rank, world
        self.voices_per_rank = voices_per_rank
        self.total_voices = voices_per_rank * world
        self.max_frames = max_frames
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        if renderer is None:
            # round-robin shards (runs of `interleave` voices dealt out to the ranks): the allocation policy
            # sweeps the pool in index order, so a burst of note-ons lands on every GPU instead of one;
            # interleave = 0 gives contiguous ranges
            kw = (dict(shard_interleave=interleave, shard_index=rank, shard_count=world) if interleave and world > 1
                  else dict(shard_begin=rank * voices_per_rank, shard_voices=voices_per_rank))
            renderer = _synth.Synth(self.total_voices, max_frames=max_frames,
                                    device=self.device.index if self.device.index is not None else -1,
                                    block_voices=block_voices, **kw)
        self.renderer = renderer
        self.combine = combine if combine is not None else self._combine_hip
        # force_exchange: a single rank still goes through the collective (a rehearsal of the N-rank path's calls on a
        # one-GPU box: tests/test_gpu_parity.py::test_sharded_exchange_through_rccl_with_one_rank)
        self.force_exchange = force_exchange and world == 1
        exchange = world > 1 or self.force_exchange
        self.overlap = overlap and exchange
        self.reduce_to_root = reduce_to_root and exchange
        # every buffer is exchanged as whole max_frames rows (4 KiB at 1024 frames: the exchange is latency-bound), so a
        # shorter fill needs neither a temporary nor a wait; what lies behind `frames` in a row is never read
        self.partial = [torch.zeros(max_frames, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.gathered = [torch.zeros((world, max_frames), dtype=torch.float32, device=self.device) for _ in range(2)]
        self.mix = torch.zeros(max_frames, dtype=torch.float32, device=self.device)
        self._pending = None      # (work, slot, frames) of the exchange still in flight
        self._k = 0
        self._host_target = None

    # ---- events: identical stream on every rank ----
    def note_events(self, events):
        self.renderer.note_events(events)

    def note_on(self, note, velocity=1.0):
        return self.renderer.note_on(note, velocity)

    def note_off(self, note):
        self.renderer.note_off(note)

    def set_patch_bank(self, patches):
        self.renderer.set_patch_bank(patches)

    def load_patch(self, text):
        self.renderer.load_patch(text)

    # ---- one buffer ----
    def _stream_ptr(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None

    def _combine_hip(self, rows, n_rows, frames, out):
        # rows: [n_rows][max_frames]; the kernel adds whole rows (row stride = max_frames)
        _synth.sum_partials_device(rows.data_ptr(), n_rows, self.max_frames, out.data_ptr(), self._stream_ptr())

    def copy_mix_to(self, pinned_host):
        """every finished mix is also copied (asynchronously, behind its combine) into this pinned host tensor"""
        self._host_target = pinned_host

    def _finish(self, pending):
        work, slot, frames = pending
        if work is not None:
            work.wait()                      # current stream waits for the collective
        if self.rank == 0:
            if self.reduce_to_root:
                self.mix[:frames].copy_(self.partial[slot][:frames])      # reduce(SUM) left the total in rank 0's row
            else:
                self.combine(self.gathered[slot], self.world, frames, self.mix)
            if self._host_target is not None:
                self._host_target.copy_(self.mix, non_blocking=True)

    def fill(self, frames, sample_rate=48000):
        """Render one buffer.  With overlap on, the exchange of buffer k runs on the collective's own
        stream while buffer k+1 renders; call flush() before reading ``mix``."""
        if self.devices:
            import numpy as np
            if self._out is None or self._out.size != frames:
                self._out = np.empty(frames, dtype=np.float32)
            self.renderer.sample(self._out, sample_rate)
            if self._host_target is not None:
                self._host_target[:frames].copy_(torch.from_numpy(self._out))
            return
        slot = self._k & 1
        self._k += 1
        part = self.partial[slot]
        if self.world == 1 and not self.force_exchange and hasattr(self.renderer, "fill_device_root") and self.combine == self._combine_hip:
            # one shard: its mix kernel roots the sum itself ((+0.0) + total), no combine pass
            self.renderer.fill_device_root(self.mix.data_ptr(), frames, sample_rate, self._stream_ptr())
            if self._host_target is not None:
                self._host_target.copy_(self.mix, non_blocking=True)
            return
        self.renderer.fill_device(part.data_ptr(), frames, sample_rate, self._stream_ptr())
        if self.world == 1 and not self.force_exchange:
            self.gathered[slot][0].copy_(part)
            self.combine(self.gathered[slot], 1, frames, self.mix)
            return
        if self._pending is not None:
            self._finish(self._pending)
            self._pending = None
        if self.device.type == "cuda" and dist.get_backend() == "gloo":
            # rehearsal of the multi-rank flow without RCCL (ranks sharing one card): the rows go through the host
            host = part.cpu()
            if self.reduce_to_root:
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                if self.rank == 0:
                    part.copy_(host)
            else:
                rows = torch.empty((self.world, self.max_frames), dtype=torch.float32)
                dist.all_gather_into_tensor(rows.view(-1), host)
                self.gathered[slot].copy_(rows)
            self._finish((None, slot, frames))
            return
        if self.reduce_to_root:
            work = dist.reduce(part, dst=0, op=dist.ReduceOp.SUM, async_op=self.overlap)
        else:
            work = dist.all_gather_into_tensor(self.gathered[slot].view(-1), part, async_op=self.overlap)
        pending = (work if self.overlap else None, slot, frames)
        if self.overlap:
            self._pending = pending
        else:
            self._finish(pending)

    def flush(self):
        if self.devices:
            return
        if self._pending is not None:
            self._finish(self._pending)
            self._pending = None
