"""Sample store of the RWR baseline on the LOCAL filesystem.

The reference writes the sampler's output through `utils.RemoteWriter` (HDF5 shards uploaded to a GCS bucket,
/root/reference/ddpo/utils/hdf5.py:72-300) and reads it back through `RemoteReader` / `H5Reader` + `BucketDataset` +
`get_bucket_loader` (/root/reference/ddpo/datasets/bucket.py).  Neither h5py nor a bucket exists here; the same roles and call
surface are kept on plain `.npz` shards in a directory:

  LocalWriter(savepath, split_size)            configure(field, encode_fn=, decode_fn=) / add_batch(batch, mask=) -> n added / close()
  LocalReader(loadpath)                        len / reader[i] -> {field: value} (+ "weights" once make_weights ran) /
                                               make_weights(field, temperature, by_prompt)   (hdf5.py:437-461: softmax_ref * N, or per prompt)
  BucketDataset, collate_fn, get_bucket_loader the dataset / loader of datasets/bucket.py: caption choice, tokenisation, uncond ids,
                                               drop_last batching in storage order, shard() over ranks
  Percentile / StreamingPercentile / Threshold / make_masker, StreamingAverage, softmax_ref       (utils/logger.py:32-94, utils/array.py:32-41)
Images are stored JPEG-encoded (`encode_jpeg`, quality 95) exactly as the reference configures its `images` field."""
import glob
import io
import json
import os
import random

import numpy as np
from PIL import Image


# ------------------------------------------------------------------------------------------------ small helpers of utils/
def softmax_ref(x, temperature=1.0):
    assert x.ndim == 1
    x = x * temperature
    z = x - x.max()
    numer = np.exp(z)
    return numer / numer.sum()


class StreamingAverage:
    def __init__(self):
        self.n, self.avg = 0, 0

    def __call__(self, x):
        self.n += 1
        self.avg = self.avg * (self.n - 1) / self.n + x / self.n


class Masker:
    def __repr__(self):
        return f"[ {self._name} | {self.p} ]"

    def mask(self, xs):
        return xs >= self.p


class Percentile(Masker):
    def __init__(self, q=90, maxsize=5e6):
        self.q, self._name = q, f"percentile: {q}"

    def __call__(self, xs):
        if xs.ndim == 2:
            xs = xs.squeeze(axis=-1)
        self.p = np.percentile(xs, self.q)
        return super().mask(xs)


class StreamingPercentile(Masker):
    def __init__(self, q=90, maxsize=5e6):
        self.q, self.xs, self.size, self._name = q, np.zeros(int(maxsize)), 0, f"streaming_percentile: {q}"

    def __call__(self, xs):
        if xs.ndim == 2:
            xs = xs.squeeze(axis=-1)
        n = len(xs)
        self.xs[self.size:self.size + n] = xs[:]
        self.size += n
        self.p = np.percentile(self.xs[:self.size], self.q)
        return super().mask(xs)


class Threshold(Masker):
    def __init__(self, threshold=0.95):
        self.p, self._name = threshold, f"threshold: {threshold}"

    def __call__(self, xs):
        return super().mask(xs)


def make_masker(mode, param):
    return {"percentile": Percentile, "streaming_percentile": StreamingPercentile, "threshold": Threshold}[mode](param)


def decode_jpeg(buf):
    return np.asarray(Image.open(io.BytesIO(np.asarray(buf, dtype=np.uint8).tobytes())))


# ------------------------------------------------------------------------------------------------ writer / reader
class LocalWriter:
    """add_batch(batch, mask) appends the masked rows field by field; every `split_size` rows a shard `<rank>_<run_id>_<i>.npz` is closed.

    Nothing that exists in `savepath` is touched before close(): shard names carry the run id (the reference's RemoteWriter uses timestamped
    names and never deletes), close() replaces this rank's manifest atomically and only THEN removes this rank's shards the new manifest does
    not list — a sampling run that crashes, or was started by mistake, leaves the previous dataset readable.  `run_id` must be the same on
    every rank of one sampling run (pipeline/sample.py broadcasts rank 0's); LocalReader refuses manifests whose ids differ."""

    def __init__(self, savepath, split_size=1000, rank=0, run_id=None):
        self.savepath, self.split_size, self.rank = savepath, int(split_size), int(rank)
        self.run_id = str(run_id) if run_id is not None else new_run_id()
        os.makedirs(savepath, exist_ok=True)
        self._encode, self._rows, self._shard, self._total = {}, [], 0, 0
        self._files = []

    def configure(self, field, max_size=None, vlen=False, encode_fn=None, decode_fn=None):
        self._encode[field] = encode_fn

    def __len__(self):
        return self._total

    def add_batch(self, batch, mask=None):
        sizes = [len(v) for v in batch.values()]
        assert len(set(sizes)) == 1, f"Batch sizes must be equal, got {sizes}"
        idx = range(sizes[0]) if mask is None else np.where(np.asarray(mask).reshape(-1))[0]
        for i in idx:
            row = {}
            for k, v in batch.items():
                x = v[i]
                enc = self._encode.get(k)
                row[k] = enc(x) if enc is not None else x
            self._rows.append(row)
            self._total += 1
            if len(self._rows) >= self.split_size:
                self._flush()
        return len(idx)

    def _flush(self):
        if not self._rows:
            return
        cols = {}
        for k in self._rows[0]:
            vals = [r[k] for r in self._rows]
            arr = np.empty(len(vals), dtype=object)
            arr[:] = vals
            cols[k] = arr
        name = f"{self.rank}_{self.run_id}_{self._shard:05d}.npz"
        np.savez(os.path.join(self.savepath, name), **cols)
        self._files.append({"file": name, "rows": len(self._rows)})
        self._rows, self._shard = [], self._shard + 1

    def close(self, metadata=None, world=1):
        """Flush, replace this rank's manifest (its shard files and row counts, the run id; `world` = number of ranks writing into this
        directory) atomically, remove this rank's shards of earlier runs (those the new manifest does not list) and, on rank 0, write
        metadata.json."""
        self._flush()
        path = os.path.join(self.savepath, f"manifest_{self.rank}.json")
        with open(path + ".tmp", "w") as f:
            json.dump({"rank": self.rank, "world": int(world), "run_id": self.run_id, "n_samples": self._total, "shards": self._files}, f, indent=2)
        os.replace(path + ".tmp", path)
        keep = {sh["file"] for sh in self._files}
        for old in glob.glob(os.path.join(self.savepath, f"{self.rank}_*.npz")):
            if os.path.basename(old) not in keep:
                os.remove(old)
        if self.rank == 0:
            # ranks >= world belong to an earlier, wider run: nobody of THIS run owns their files, so rank 0 removes them (they
            # piled up across reruns with a smaller world; the reader already ignored them)
            import re
            for old in glob.glob(os.path.join(self.savepath, "manifest_*.json")) + glob.glob(os.path.join(self.savepath, "*_*.npz")):
                m = re.match(r"(?:manifest_)?(\d+)[_.]", os.path.basename(old))
                if m and int(m.group(1)) >= int(world):
                    os.remove(old)
        if metadata is not None and self.rank == 0:
            with open(os.path.join(self.savepath, "metadata.json.tmp"), "w") as f:
                json.dump(metadata, f, indent=2, default=str)
            os.replace(os.path.join(self.savepath, "metadata.json.tmp"), os.path.join(self.savepath, "metadata.json"))


def new_run_id():
    """Identifier of one sampling run (timestamp + random suffix): part of every shard name and manifest of the run."""
    import time
    import uuid
    return time.strftime("%Y%m%d-%H%M%S") + "-" + uuid.uuid4().hex[:8]


class LocalReader:
    def __init__(self, loadpath):
        """Reads exactly the shards the writers' manifests list (manifest_<rank>.json, one per rank of the sampling run: a stale shard of an
        earlier run with more ranks / shards is ignored, a missing listed shard or a missing rank's manifest is an error); a directory without
        manifests (written before round 4) falls back to every *.npz."""
        manifests = sorted(glob.glob(os.path.join(loadpath, "manifest_*.json")))
        if manifests:
            metas = [json.load(open(m)) for m in manifests]
            r0 = [m for m in metas if int(m["rank"]) == 0]
            if not r0:
                raise FileNotFoundError(f"'{loadpath}': rank 0's manifest is missing")
            world = int(r0[0].get("world", 1))           # the run that wrote LAST is rank 0's: a stale manifest of a wider earlier run is ignored
            ranks = sorted(int(m["rank"]) for m in metas if int(m["rank"]) < world)
            if ranks != list(range(world)):
                raise FileNotFoundError(f"'{loadpath}': manifests of ranks {ranks} found, the sampling run had {world} ranks")
            # every rank of ONE run carries the same run id: a stale rank-0 manifest next to newer shards of the other ranks (or the other way
            # round: a run that died before every rank closed) is refused instead of being read as a mixture of two runs
            ids = {int(m["rank"]): m.get("run_id") for m in metas if int(m["rank"]) < world}
            if any(v is None for v in ids.values()):
                # manifests written before run ids existed (round 4) all read as None and would pass the equality check below: say so
                import warnings
                warnings.warn(f"'{loadpath}': manifests without a run id (written before round 5) — cannot tell whether the ranks' shards belong "
                              "to one sampling run")
            if len(set(ids.values())) != 1:
                raise FileNotFoundError(f"'{loadpath}': the ranks' manifests belong to different sampling runs (run ids {ids})")
            files = [os.path.join(loadpath, sh["file"]) for m in sorted(metas, key=lambda m: int(m["rank"])) if int(m["rank"]) < world for sh in m["shards"]]
            missing = [f for f in files if not os.path.exists(f)]
            if missing:
                raise FileNotFoundError(f"'{loadpath}': shards listed in the manifest are missing: {missing[:3]}")
        else:
            files = sorted(glob.glob(os.path.join(loadpath, "*.npz")))
        if not files:
            raise FileNotFoundError(f"no sample shards (*.npz) in '{loadpath}'")
        self._cols = {}
        for f in files:
            with np.load(f, allow_pickle=True) as z:
                for k in z.files:
                    self._cols.setdefault(k, []).extend(list(z[k]))
        self._keys = list(self._cols)
        self.weighted, self.weights = False, None

    def __len__(self):
        return len(self._cols[self._keys[0]])

    def get(self, idx, field):
        v = self._cols[field][idx]
        return np.stack([np.asarray(x) for x in v]) if isinstance(idx, slice) else v

    def __getitem__(self, idx):
        batch = {k: self._cols[k][idx] for k in self._keys}
        if self.weighted:
            batch["weights"] = self.weights[idx]
        return batch

    def make_weights(self, field, temperature, by_prompt):
        labels = np.asarray(self.get(slice(0, len(self)), field), dtype=np.float64).squeeze()
        if by_prompt:
            prompts = np.asarray(self.get(slice(0, len(self)), "inference_prompts")).squeeze()
            self.weights = np.empty_like(labels)
            for prompt in np.unique(prompts):
                mask = prompts == prompt
                self.weights[mask] = softmax_ref(labels[mask], temperature=temperature) * mask.sum()
        else:
            self.weights = softmax_ref(labels, temperature=temperature) * len(self)
        self.weighted = True
        cumsum = np.cumsum(np.sort(self.weights)[::-1] / len(self))
        n = ((cumsum <= 0.9) * np.arange(len(cumsum))).max()
        print(f"[ utils/bucket ] Weights sanity check: {n} / {len(cumsum)} ({(n / len(cumsum)):.3}%) samples account for 90% of the weight | "
              f"temperature: {temperature}")


# ------------------------------------------------------------------------------------------------ dataset / loader (datasets/bucket.py)
class BucketDataset:
    def __init__(self, reader):
        self.reader = reader
        self._max_size, self._offset = None, 0
        self._shuffled = np.arange(len(self))

    def __len__(self):
        return self._max_size or len(self.reader)

    def __getitem__(self, idx):
        worker_idx = self._offset + idx
        shuffled_idx = int(self._shuffled[worker_idx])
        x = dict(self.reader[shuffled_idx])
        caption = x["training_prompts"]
        if isinstance(caption, (list, np.ndarray)):
            caption = random.choice(list(caption))          # select_caption
        x["text"] = caption
        x.update(idx=worker_idx, shuffled_idx=shuffled_idx)
        return x

    def shuffle(self):
        self._shuffled = np.random.permutation(self._shuffled)

    def shard(self, host_id=0, n_hosts=1):
        n = len(self) // n_hosts
        self._max_size, self._offset = n, host_id * n

    def make_weights(self, *a, **kw):
        self.reader.make_weights(*a, **kw)

    def subsample(self, N):
        self._max_size = N


def collate_fn(tokenizer, examples, image_field="vae", text_field="input_ids"):
    pixel_values = np.stack([np.asarray(e[image_field]) for e in examples]).astype(np.float32)
    captions = [str(e["text"]) for e in examples]
    labels = {k: np.stack([np.asarray(e[k]) for e in examples]) for k in ("aesthetic", "consistency", "jpeg", "neg_jpeg", "labels", "weights")
              if k in examples[0]}
    tok = lambda texts: tokenizer(texts, padding="max_length", max_length=tokenizer.model_max_length, return_tensors="np").input_ids
    return {image_field: pixel_values, text_field: tok(captions), "idxs": np.stack([e["idx"] for e in examples]),
            "shuffled_idxs": np.stack([e["shuffled_idx"] for e in examples]), "uncond_text": tok([""] * len(examples)), **labels}


class _Loader:
    """torch DataLoader(shuffle=False, drop_last=True) of the reference, without worker processes."""

    def __init__(self, dataset, tokenizer, batch_size):
        self.dataset, self.tokenizer, self.batch_size = dataset, tokenizer, int(batch_size)

    def __len__(self):
        return len(self.dataset) // self.batch_size

    def __iter__(self):
        for b in range(len(self)):
            yield collate_fn(self.tokenizer, [self.dataset[b * self.batch_size + i] for i in range(self.batch_size)])


def get_bucket_loader(loadpath, tokenizer, batch_size, resolution=None, max_train_samples=None, host_id=0, n_hosts=1):
    dataset = BucketDataset(LocalReader(loadpath))
    if max_train_samples is not None:
        dataset.subsample(max_train_samples)
    dataset.shard(host_id, n_hosts)
    return dataset, _Loader(dataset, tokenizer, batch_size)
