"""Utterance tables, batch assembly and host->HBM staging (reference: glow_tts_train/dataset.py).

Same surface as the reference — `PhonemeMelLoader`, `PhonemeMelCollate`, `load_phonemes`, `load_mels` — and the same
batch tuple `(text_padded, input_lengths, mel_padded, output_lengths, speaker_ids)`, plus what an MI355X training loop
needs around it:

* `PhonemeMelCollate(pin_memory=True)` assembles each batch straight into page-locked staging slots (a small ring), so
  the host->device copy is one DMA per tensor with no pageable bounce;
* `DeviceBatches` walks any loader one batch ahead: batch k+1 is copied on its own stream while step k computes, and
  the consumer stream only waits on that copy's event (the reference's `to_gpu` calls sit at the head of every step,
  train.py:107-111).

Nothing here touches the HIP library; tensors stay on the host until `DeviceBatches` moves them.
"""
from __future__ import annotations

import csv
import json
import logging
import random
import typing
from pathlib import Path

import numpy as np
import torch
import torch.utils.data

_LOGGER = logging.getLogger("glow_tts_train.dataset")

UttKey = typing.Tuple[int, str]


class PhonemeMelLoader(torch.utils.data.Dataset):
    """(speaker, utterance id) -> (phoneme ids, mel (n_mels, T)[, speaker]) (reference dataset.py:20-70).

    Mels missing from `id_mels` are read from `<mel_dirs[speaker]>/<utt_id>.npy` on first use and kept."""

    def __init__(self, id_phonemes: typing.Dict[UttKey, torch.Tensor], id_mels: typing.Dict[UttKey, torch.Tensor],
                 mel_dirs: typing.Optional[typing.Dict[int, Path]] = None, multispeaker: bool = False):
        self.id_phonemes, self.id_mels, self.mel_dirs, self.multispeaker = id_phonemes, id_mels, mel_dirs, multispeaker
        if id_mels:
            self.ids = [k for k in id_phonemes if k in id_mels]
            assert self.ids, "No shared utterance ids between phonemes and mels"
        else:
            self.ids = list(id_phonemes)                 # every mel is expected under mel_dirs
        random.shuffle(self.ids)

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, index):
        key = self.ids[index]
        speaker, utt_id = key
        mel = self.id_mels.get(key)
        if mel is None:
            mel_dir = (self.mel_dirs or {}).get(speaker)
            assert mel_dir, f"Missing mel for id {utt_id}, but no mels_dir"
            mel = torch.from_numpy(np.load(Path(mel_dir) / (utt_id + ".npy"), allow_pickle=True))
            self.id_mels[key] = mel
        text = self.id_phonemes[key]
        return (text, mel, len(text), speaker) if self.multispeaker else (text, mel, len(text))


class _StagingRing:
    """`slots` page-locked byte arenas handed out round-robin; a slot is rewritten `slots` batches later."""

    def __init__(self, slots: int):
        self.bufs: typing.List[typing.Optional[torch.Tensor]] = [None] * slots
        self.k = 0

    def take(self, nbytes: int) -> torch.Tensor:
        i, self.k = self.k, (self.k + 1) % len(self.bufs)
        buf = self.bufs[i]
        if buf is None or buf.numel() < nbytes:
            buf = self.bufs[i] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, pin_memory=True)
        return buf


def _carve(arena: typing.Optional[torch.Tensor], off: int, shape, dtype) -> typing.Tuple[torch.Tensor, int]:
    n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    if arena is None:
        return torch.zeros(shape, dtype=dtype), off
    t = arena[off:off + n].view(dtype).view(shape)
    t.zero_()
    return t, off + ((n + 63) & ~63)


class PhonemeMelCollate:
    """Pad one list of utterances into a batch (reference dataset.py:72-116).

    Rows are ordered by text length, longest first (`torch.sort(..., descending=True)`, the order the reference's
    alignment masks assume); text is right-padded with id 0 to the longest text, mels with 0.0 to the longest mel
    rounded up to a multiple of `n_frames_per_step`; `speaker_ids` is None unless `multispeaker`.
    With `pin_memory=True` the returned tensors are views of a page-locked staging slot that is reused `slots`
    batches later — consume (or copy) a batch before collating `slots` more."""

    def __init__(self, n_frames_per_step: int = 1, multispeaker: bool = False, pin_memory: bool = False, slots: int = 3):
        self.n_frames_per_step, self.multispeaker = n_frames_per_step, multispeaker
        self._ring = _StagingRing(slots) if pin_memory else None

    def __call__(self, batch):
        n = len(batch)
        sorted_lengths, order = torch.sort(torch.tensor([len(item[0]) for item in batch], dtype=torch.long), dim=0,
                                           descending=True)
        order = order.tolist()
        t_text = int(sorted_lengths[0])
        n_mels = batch[0][1].size(0)
        t_mel = max(item[1].size(1) for item in batch)
        t_mel += -t_mel % self.n_frames_per_step

        arena = None
        if self._ring is not None:
            arena = self._ring.take(8 * n * t_text + 4 * n * n_mels * t_mel + 24 * n + 5 * 64)
        text_padded, off = _carve(arena, 0, (n, t_text), torch.long)
        input_lengths, off = _carve(arena, off, (n,), torch.long)
        input_lengths.copy_(sorted_lengths)
        mel_padded, off = _carve(arena, off, (n, n_mels, t_mel), torch.float32)
        output_lengths, off = _carve(arena, off, (n,), torch.long)
        speaker_ids = None
        if self.multispeaker:
            speaker_ids, off = _carve(arena, off, (n,), torch.long)
        for row, src in enumerate(order):
            item = batch[src]
            text, mel = item[0], item[1]
            text_padded[row, :text.size(0)] = text
            mel_padded[row, :, :mel.size(1)] = mel
            output_lengths[row] = mel.size(1)
            if speaker_ids is not None:
                speaker_ids[row] = item[3]
        return text_padded, input_lengths, mel_padded, output_lengths, speaker_ids


class DeviceBatches:
    """Iterate a loader's collated batches as device tensors, staying `depth` batches ahead of the consumer.

    The copies run on a dedicated stream; a batch is yielded after the consumer's current stream has been told to wait
    for that batch's copy event (no host synchronisation on the consumer side).  Before the loader is asked for batch
    k+depth+1 the copy of batch k is known to have finished, so a collate with `slots >= depth + 1` staging slots never
    rewrites memory a DMA still reads.  Tensors are `record_stream`-ed on the consumer stream, so the caching allocator
    will not hand their memory to the copy stream while the step still uses it."""

    def __init__(self, loader: typing.Iterable, device=None, depth: int = 1):
        self.loader, self.depth = loader, max(1, depth)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceBatches stages batches into HBM; it needs a GPU device")
        self.copy_stream = torch.cuda.Stream(self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        with torch.cuda.stream(self.copy_stream):
            moved = tuple(None if t is None else t.to(self.device, non_blocking=True) for t in batch)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        return moved, done

    def __iter__(self):
        queue: typing.List[typing.Tuple[tuple, torch.cuda.Event]] = []
        retired: typing.Optional[torch.cuda.Event] = None
        source = iter(self.loader)
        exhausted = False
        while True:
            while not exhausted and len(queue) < self.depth + 1:
                if retired is not None:
                    retired.synchronize()        # the slot the next collate may rewrite has been read
                    retired = None
                try:
                    queue.append(self._stage(next(source)))
                except StopIteration:
                    exhausted = True
            if not queue:
                return
            moved, done = queue.pop(0)
            consumer = torch.cuda.current_stream(self.device)
            consumer.wait_event(done)
            for t in moved:
                if t is not None:
                    t.record_stream(consumer)
            retired = done
            yield moved


def load_phonemes(csv_file: typing.TextIO, config) -> typing.Dict[str, torch.Tensor]:
    """`utt_id|p1 p2 p3 ...` rows -> {utt_id: IntTensor}; rows outside [min_seq_length, max_seq_length] are dropped
    (reference dataset.py:122-162)."""
    lo, hi = config.min_seq_length, config.max_seq_length
    phonemes: typing.Dict[str, torch.Tensor] = {}
    short = long_ = 0
    for row in csv.reader(csv_file, delimiter="|"):
        utt_id, ids = row[0], [int(p) for p in row[1].strip().split()]
        if lo is not None and len(ids) < lo:
            short += 1
            continue
        if hi is not None and len(ids) > hi:
            long_ += 1
            continue
        phonemes[utt_id] = torch.tensor(ids, dtype=torch.int32)
    if short or long_:
        _LOGGER.warning("Dropped some utterance (%s too small, %s too large)", short, long_)
    return phonemes


def load_mels(jsonl_file: typing.TextIO) -> typing.Dict[str, torch.Tensor]:
    """One JSON object per line, `{"id": ..., "mel": [[...], ...]}` -> {id: FloatTensor (n_mels, T)}
    (reference dataset.py:165-176)."""
    mels: typing.Dict[str, torch.Tensor] = {}
    for line in jsonl_file:
        line = line.strip()
        if line:
            obj = json.loads(line)
            mels[obj["id"]] = torch.tensor(obj["mel"], dtype=torch.float32)
    return mels
