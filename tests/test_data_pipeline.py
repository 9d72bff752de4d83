"""Input pipeline (reference dataset.py:11-134): YAML pair list -> GRAY decode -> antialiased bilinear
resize to 256 x 256 on uint8 -> float32 / 255 -> [-1, 1]; sharding of the lists across data-parallel ranks
(what Lightning's DDP strategy does to the reference's loaders, reference main.py:123-136).  CPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_pairs(tmp_path, n, size=256, seed=0):
    """n (input, ground_truth) PNG pairs + the YAML list the reference's README describes (README.md:35-48)."""
    from PIL import Image
    import yaml
    rng = np.random.default_rng(seed)
    items, arrays = [], []
    os.makedirs(tmp_path / "img", exist_ok=True)
    for i in range(n):
        a = rng.integers(0, 256, (size, size), dtype=np.uint8)
        b = rng.integers(0, 256, (size, size), dtype=np.uint8)
        Image.fromarray(a, mode="L").save(tmp_path / "img" / f"in_{i:03d}.png")
        Image.fromarray(b, mode="L").save(tmp_path / "img" / f"gt_{i:03d}.png")
        items.append({"input": f"img/in_{i:03d}.png", "ground_truth": f"img/gt_{i:03d}.png"})
        arrays.append((a, b))
    with open(tmp_path / "list.yaml", "w") as f:
        yaml.safe_dump(items, f)
    return tmp_path / "list.yaml", arrays


def test_image_datamodule_values_shape_range(pai, tmp_path):
    from thesis_pai_reconstruction_amd.dataset import ImageDataModule
    lst, arrays = _write_pairs(tmp_path, 5)
    dm = ImageDataModule(str(lst), str(lst), batch_size=2, normalize=True, world=1, rank=0)
    dm.setup("fit")
    seen = 0
    for x, t in dm.val_dataloader():                      # un-shuffled: list order
        assert x.dtype == torch.float32 and x.shape[1:] == (1, 256, 256) and t.shape == x.shape
        assert float(x.min()) >= -1.0 and float(x.max()) <= 1.0
        for k in range(x.shape[0]):
            a, b = arrays[seen + k]
            # 256 x 256 sources are not resized: exactly uint8 / 255 * 2 - 1 (reference dataset.py:53-59, Q2-corrected)
            assert torch.equal(x[k, 0], torch.from_numpy(a).float().div(255) * 2 - 1)
            assert torch.equal(t[k, 0], torch.from_numpy(b).float().div(255) * 2 - 1)
        seen += x.shape[0]
    assert seen == 5
    batches = list(dm.train_dataloader())
    assert sum(b[0].shape[0] for b in batches) == 5 and batches[-1][0].shape[0] == 1      # drop_last=False
    dm0 = ImageDataModule(str(lst), None, batch_size=2, normalize=False, world=1, rank=0)
    dm0.setup("predict")
    x, _ = next(iter(dm0.predict_dataloader()))
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0


@pytest.mark.parametrize("shape", [(512, 512), (300, 400), (128, 128), (257, 255)])
def test_resize_matches_antialiased_bilinear(pai, tmp_path, shape):
    """The reference resizes with torchvision Resize((256, 256), antialias=True) on the uint8 tensor (dataset.py:51-54):
    torchvision 0.15 = cast to float32, aten's antialiased bilinear kernel, round, cast back to uint8.  load_gray_256
    must give the same BYTES (round 2 used PIL's BILINEAR filter and was one uint8 step off on some pixels; that
    filter is kept here only as a sanity bound on the reference expression itself)."""
    from PIL import Image
    from thesis_pai_reconstruction_amd.dataset import load_gray_256
    rng = np.random.default_rng(7)
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w]
    img = (127 + 80 * np.sin(yy / 7.0) * np.cos(xx / 5.0) + rng.normal(0, 20, (h, w))).clip(0, 255).astype(np.uint8)
    path = tmp_path / "a.png"
    Image.fromarray(img, mode="L").save(path)
    got = load_gray_256(str(path), 256, normalize=False)                 # [1, 256, 256] in [0, 1]
    assert got.shape == (1, 256, 256)
    got_u8 = (got * 255).round().to(torch.int32)
    ref = F.interpolate(torch.from_numpy(img)[None, None].float(), size=(256, 256), mode="bilinear",
                        antialias=True, align_corners=False)
    ref_u8 = ref.round().clamp(0, 255).to(torch.int32)[0]
    diff = (got_u8 - ref_u8).abs()
    assert int(diff.max()) == 0, int(diff.max())
    # what is stored is exactly k / 255 in fp32, k the byte: the byte is recoverable without rounding slack
    assert torch.equal(got, ref_u8.to(torch.float32).div(255))
    # independent sanity bound on the reference expression: PIL's fixed-point BILINEAR filter is within one step of it
    pil = torch.from_numpy(np.asarray(Image.open(path).convert("L").resize((256, 256), Image.BILINEAR), dtype=np.uint8).copy()).to(torch.int32)
    if shape != (256, 256):
        assert int((pil - ref_u8[0]).abs().max()) <= 1


def test_sharded_loader_partitions_the_list(pai):
    from thesis_pai_reconstruction_amd.dataset import ShardedLoader

    class Idx(torch.utils.data.Dataset):
        def __len__(self):
            return 11

        def __getitem__(self, i):
            return torch.tensor(i)

    for world in (2, 3, 4):
        per_rank = []
        for rank in range(world):
            ld = ShardedLoader(Idx(), 2, True, world, rank, seed=5)
            ld.set_epoch(3)
            per_rank.append(torch.cat([b for b in ld]).tolist())
        assert len({len(r) for r in per_rank}) == 1                       # same number of samples on every rank
        allidx = sum(per_rank, [])
        assert set(allidx) == set(range(11))                              # one pass covers the data ...
        assert len(allidx) - 11 < world                                   # ... padded by wrap-around only
        ld = ShardedLoader(Idx(), 2, True, world, 0, seed=5)
        ld.set_epoch(4)
        assert torch.cat([b for b in ld]).tolist() != per_rank[0]         # reshuffled per epoch
    one = ShardedLoader(Idx(), 4, False, 1, 0)
    assert torch.cat([b for b in one]).tolist() == list(range(11)) and one.sampler is None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, list_file, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import pai_bootstrap
    pai_bootstrap.load()
    import torch.distributed as dist
    from thesis_pai_reconstruction_amd import dist as pdist
    from thesis_pai_reconstruction_amd.dataset import ImageDataModule
    pdist.init_from_env(backend="gloo")
    try:
        dm = ImageDataModule(list_file, list_file, batch_size=2)          # world / rank from torch.distributed
        assert (dm.world, dm.rank) == (world, rank)
        dm.setup("fit")
        out = {}
        for name, ld in (("train", dm.train_dataloader()), ("val", dm.val_dataloader())):
            ld.set_epoch(0)
            # identify each sample by its first pixel row (unique per random image)
            out[name] = [tuple(x[k, 0, 0, :8].tolist()) for x, _ in ld for k in range(x.shape[0])]
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_shards_partition_the_dataset(pai, tmp_path):
    lst, arrays = _write_pairs(tmp_path, 7, seed=3)
    keys = {tuple((torch.from_numpy(a[0, :8].copy()).float().div(255) * 2 - 1).tolist()): i
            for i, (a, _) in enumerate(arrays)}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, str(lst), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for split in ("train", "val"):
        ids = [[keys[k] for k in res[r][split]] for r in range(2)]
        assert len(ids[0]) == len(ids[1]) == 4                            # ceil(7 / 2) each
        assert set(ids[0]) | set(ids[1]) == set(range(7))                 # the shards cover the list
        assert len(set(ids[0]) & set(ids[1])) <= 1                        # at most the wrap-around pad is shared
    assert [keys[k] for k in res[0]["val"]] == [0, 2, 4, 6]               # un-shuffled validation: strided shard
