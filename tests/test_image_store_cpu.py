"""Host side of the image input path, no GPU: the oracle's ToTensor restatement, the decode fixture F14, ImageRef items
of the pair dataset, and the negative lookahead's stream order."""
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN
from oracle import cone_oracle as O

from learning_embeddings_amd.image_store import ImageRef, StoreView, decode_u8
from learning_embeddings_amd.oe_h_trainer import DiGraph, ETHECHierarchyWithImages, RandomHorizontalFlip
from learning_embeddings_amd.hierarchy import NegativeGraph, SyntheticLabelMap
from learning_embeddings_amd import parallel


def test_oracle_image_batch_is_totensor_on_every_byte_value():
    """ToTensor is `img.to(float32).div(255)` on the HWC -> CHW permuted uint8 array (torchvision functional.to_tensor; torchvision is
    not installed: the same two torch ops stand in for it).  Every byte value, both flips, c_out 3 and 4."""
    u8 = np.arange(256, dtype=np.uint8).reshape(1, 2, 128, 1).repeat(3, axis=3).copy()
    u8[..., 1] = 255 - u8[..., 1]; u8[..., 2] //= 2
    u8 = np.concatenate([u8, u8[:, ::-1].copy()])
    ref = torch.from_numpy(u8).permute(0, 3, 1, 2).contiguous().to(torch.float32).div(255)
    got = O.image_batch(u8)
    assert got.dtype == np.float32 and np.array_equal(got, ref.numpy())
    flipped = O.image_batch(u8, flips=[1, 0], c_out=4)
    assert np.array_equal(flipped[0, :3], ref[0].flip(-1).numpy()) and np.array_equal(flipped[1, :3], ref[1].numpy())
    assert not flipped[:, 3].any()


@pytest.mark.parametrize('ext', ['png', 'jpg'])
def test_decode_u8_matches_the_committed_fixture_f14(ext):
    """decode_u8 = decode + `Resize((224, 224))` (PIL bilinear) in the reference's B, G, R order (oe_h.py:668-677, 700-712).  Expected arrays
    were produced by PIL in the build container (make_golden_images.py; cv2 is not installed -- see that script's header for the JPEG gap)."""
    z = np.load(os.path.join(GOLDEN, 'F14_image_decode.npz'))
    got = decode_u8(os.path.join(GOLDEN, 'images', 'picture.' + ext), 224)
    assert got.dtype == np.uint8 and got.shape == (224, 224, 3) and got.flags['C_CONTIGUOUS']
    assert np.array_equal(got, z['resized_bgr_' + ext])
    if ext == 'png':                                                    # lossless: the decoder's output IS the picture
        assert np.array_equal(z['decoded_rgb_png'], z['rgb'])
    else:                                                               # the JPEG stays within a few grey levels of the source picture
        assert np.abs(z['decoded_rgb_jpg'].astype(int) - z['rgb'].astype(int)).mean() < 6.0
    # channel order: the picture's RED gradient runs along x; in the B, G, R array it must be channel 2
    assert got[:, -1, 2].mean() > got[:, 0, 2].mean() + 150


def _tiny_dataset(tmp_path, transform, n=6):
    from PIL import Image
    lm = SyntheticLabelMap([2, 4])
    G = DiGraph()
    locs, names = {}, []
    r = np.random.RandomState(0)
    for j in range(n):
        nm = 'img_%d' % j
        p = os.path.join(str(tmp_path), nm + '.png')
        Image.fromarray(r.randint(0, 256, size=(40 + j, 50, 3)).astype(np.uint8)).save(p)
        locs[nm] = p; names.append(nm)
        G.add_edge(2 + j % 4, nm)
    ds = ETHECHierarchyWithImages(G, lm, imageless_dataloaders=[{'image_filename': names, 'path_to_image': [locs[n_] for n_ in names]}],
                                  transform=transform)
    return ds, locs, names


def test_dataset_items_travel_as_image_refs_and_consume_the_same_rng(tmp_path):
    """With a store view, a file-backed item is an ImageRef (name + the train transform's coin); without, the reference's float tensor.  Both
    modes draw the flip from torch's RNG once per image, so the same seed mirrors the same items; the ref's pixels + flip reproduce the tensor."""
    ds, locs, names = _tiny_dataset(tmp_path, RandomHorizontalFlip(0.5))
    torch.manual_seed(11)
    tensors = [ds[i]['to'] for i in range(len(ds))]
    flags = torch.zeros(len(names), dtype=torch.uint8)
    ds.store_view = StoreView({n: i for i, n in enumerate(names)}, flags, 224)
    torch.manual_seed(11)
    refs = [ds[i]['to'] for i in range(len(ds))]
    assert all(isinstance(r, ImageRef) and r.pixels is None for r in refs)       # main process: the store's pool decodes, not __getitem__
    assert [r.name for r in refs] == [ds.edge_of(i)[1] for i in range(len(ds))]
    assert any(r.flip for r in refs) and not all(r.flip for r in refs)
    for r, t in zip(refs, tensors):
        u8 = decode_u8(locs[r.name], 224)[None]
        assert np.array_equal(O.image_batch(u8, [r.flip])[0], t.numpy())
    # get_image stays the reference's API: a float tensor, never mirrored (oe_h.py:668-677)
    g = ds.get_image(names[0])
    assert torch.is_tensor(g) and np.array_equal(g.numpy(), O.image_batch(decode_u8(locs[names[0]], 224)[None])[0])
    assert ds.store_view is not None
    # a transform the store cannot express (no decide()) keeps the tensor path for train items
    ds.transform = lambda img: img * 0.5
    assert torch.is_tensor(ds[0]['to'])


def _worker_items(ds):
    from learning_embeddings_amd.oe_h import my_collate
    dl = torch.utils.data.DataLoader(ds, batch_size=3, num_workers=2, collate_fn=my_collate, shuffle=False)
    return [e for b in dl for e in b['to']]


def test_dataloader_workers_decode_misses_and_skip_resident_images(tmp_path):
    ds, locs, names = _tiny_dataset(tmp_path, RandomHorizontalFlip(0.5))
    flags = torch.zeros(len(names), dtype=torch.uint8).share_memory_()
    ds.store_view = StoreView({n: i for i, n in enumerate(names)}, flags, 224)
    flags[1] = 1; flags[4] = 1                                           # "resident in HBM"
    items = _worker_items(ds)
    assert [r.name for r in items] == [ds.edge_of(i)[1] for i in range(len(ds))]
    for r in items:
        if r.name in (names[1], names[4]):
            assert r.pixels is None
        else:
            assert r.pixels.dtype == torch.uint8 and np.array_equal(r.pixels.numpy(), decode_u8(locs[r.name], 224))


def test_negative_lookahead_consumes_the_stream_in_batch_order():
    """parallel.NegativePrefetcher over a finite list of batches: same negatives, in the same order, as drawing batch after batch on the
    caller's thread (the reference's order, oe_h.py:940-957); on_item sees every shard before the consumer; the end is signalled."""
    lm = SyntheticLabelMap([2, 4, 8])
    par = lm.parents()
    edges = [(p, c) for c, ps in par.items() for p in ps]
    n_img = 40
    ptr = np.arange(n_img + 1, dtype=np.int64) * 3
    adj = []
    for j in range(n_img):
        leaf = lm.level_start[2] + j % 8; mid = par[leaf][0]; top = par[mid][0]
        adj += [top, mid, leaf]
    mk = lambda: NegativeGraph(lm.levels, edges, ptr, np.asarray(adj, dtype=np.int32), pick_per_level=True, seed=0)
    r = np.random.RandomState(1)
    batches = []
    for _ in range(5):
        img = r.randint(0, n_img, size=8)
        frm = np.asarray([adj[3 * j + r.randint(0, 3)] for j in img], dtype=np.int32)
        batches.append((frm, (lm.n_classes + img).astype(np.int32)))
    g0 = mk()
    want = [g0.draw_batch(f, t, 3) for f, t in batches]
    seen = []
    g1 = mk()
    pf = parallel.NegativePrefetcher(g1, lambda s: batches[s] if s < len(batches) else None, 3, depth=2, on_item=lambda it: seen.append(it[2].copy()),
                                     rank=0, world=1)
    for s, (f, t) in enumerate(batches):
        pf_from, pf_to, neg = pf.next()
        assert np.array_equal(pf_from, f) and np.array_equal(pf_to, t) and np.array_equal(neg, want[s])
    with pytest.raises(StopIteration):
        pf.next()
    pf.close()
    assert len(seen) == len(batches) and all(np.array_equal(a, b) for a, b in zip(seen, want))
    # data parallel: each rank draws the GLOBAL batch and keeps its shard
    g2 = mk()
    pf2 = parallel.NegativePrefetcher(g2, lambda s: batches[s] if s < 2 else None, 3, rank=1, world=2)
    for s in range(2):
        f, t, neg = pf2.next()
        assert np.array_equal(f, batches[s][0][4:]) and np.array_equal(neg, want[s][4:])
    pf2.close()
