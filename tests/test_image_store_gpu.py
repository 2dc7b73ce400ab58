"""The image input path on the MI355X: lec_image_gather_u8 through the C ABI against the oracle (bit-exact), the
HBM image store's bookkeeping, and the drop-in trainer fed from image FILES -- store-backed rows bit-equal to the reference's host tensors
(ETHECHierarchyWithImages.get_image / __getitem__), with and without DataLoader workers, with and without the negative lookahead."""
import os
import numpy as np
import pytest
import torch
from oracle import cone_oracle as O

pytestmark = pytest.mark.gpu

from learning_embeddings_amd import _lib, oe_h  # noqa: E402
from learning_embeddings_amd.hierarchy import SyntheticLabelMap  # noqa: E402
from learning_embeddings_amd.image_store import ImageStore, decode_u8  # noqa: E402

DEV = 'cuda'


@pytest.mark.parametrize('H,W', [(224, 224), (32, 36), (7, 5), (3, 4)])
@pytest.mark.parametrize('c_out', [3, 4])
def test_image_gather_u8_bit_exact_vs_oracle(H, W, c_out):
    r = np.random.RandomState(H * 100 + W)
    S = 6
    u8 = r.randint(0, 256, size=(S, H, W, 3)).astype(np.uint8)
    u8[0, 0, :, 0] = np.arange(W) % 256                                 # a ramp: a mirror that is off by one pixel shows
    slots = np.asarray([3, 0, 5, 3, 1, 0, 2], dtype=np.int32)
    flips = np.asarray([0, 1, 1, 1, 0, 0, 1], dtype=np.uint8)
    store = torch.from_numpy(u8).to(DEV)
    d_slots = torch.from_numpy(slots).to(DEV); d_flips = torch.from_numpy(flips).to(DEV)
    out = torch.full((len(slots), H, W, c_out), -1.0, device=DEV)
    _lib.check(_lib.lib.lec_image_gather_u8(_lib.dptr(store), S, _lib.dptr(d_slots), _lib.dptr(d_flips), len(slots), H, W, c_out, _lib.dptr(out),
                                            _lib.stream_ptr()))
    want = O.image_batch(u8[slots], flips, c_out)                       # [n, c_out, H, W]
    assert np.array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), want)
    # no flips (NULL), and a slot outside the store reads as a black image
    bad = torch.tensor([2, 6, -1], dtype=torch.int32, device=DEV)
    out2 = torch.full((3, H, W, c_out), -1.0, device=DEV)
    _lib.check(_lib.lib.lec_image_gather_u8(_lib.dptr(store), S, _lib.dptr(bad), None, 3, H, W, c_out, _lib.dptr(out2), _lib.stream_ptr()))
    got = out2.permute(0, 3, 1, 2).cpu().numpy()
    assert np.array_equal(got[0], O.image_batch(u8[2:3], None, c_out)[0]) and not got[1:].any()


def test_image_gather_u8_rejects_bad_arguments():
    store = torch.zeros((2, 4, 4, 3), dtype=torch.uint8, device=DEV)
    slots = torch.zeros(1, dtype=torch.int32, device=DEV)
    out = torch.zeros((1, 4, 4, 4), device=DEV)
    call = lambda *a: _lib.lib.lec_image_gather_u8(*a, _lib.stream_ptr())
    assert call(None, 2, _lib.dptr(slots), None, 1, 4, 4, 4, _lib.dptr(out)) == _lib.E_ARG
    assert call(_lib.dptr(store), 2, _lib.dptr(slots), None, 1, 4, 4, 5, _lib.dptr(out)) == _lib.E_ARG
    assert call(_lib.dptr(store), 0, _lib.dptr(slots), None, 1, 4, 4, 4, _lib.dptr(out)) == _lib.E_ARG
    assert call(_lib.dptr(store), 2, _lib.dptr(slots), None, 0, 4, 4, 4, _lib.dptr(out)) == _lib.E_ARG
    assert b'c_out' in _lib.lib.lec_last_error() or b'need' in _lib.lib.lec_last_error()


def _synthetic_decoder(calls):
    def dec(loc, hw):
        calls.append(loc)
        j = int(loc)
        r = np.random.RandomState(j)
        return r.randint(0, 256, size=(hw, hw, 3)).astype(np.uint8)
    return dec


def test_image_store_fills_on_first_touch_and_replaces_first_in_first_out():
    calls = []
    dec = _synthetic_decoder(calls)
    names = ['n%d' % j for j in range(20)]
    st = ImageStore({n: str(j) for j, n in enumerate(names)}, DEV, hw=16, capacity=8, decode_threads=3, decoder=dec)
    pix = lambda j: dec(str(j), 16)
    calls.clear()

    def check(ask, flips=None):
        got = st.batch([names[j] for j in ask], flips).cpu().numpy()
        assert np.array_equal(got, O.image_batch(np.stack([pix(j) for j in ask]), flips))

    check([0, 1, 2, 2, 0])
    assert st.stats['decoded_here'] == 3 and st.stats['uploads'] == 1 and st.stats['hits'] == 0       # ONE copy for a cold step
    assert [bool(st.flags[j]) for j in range(4)] == [True, True, True, False]
    n_dec = len([c for c in calls])
    check([2, 1], flips=[1, 0])
    assert st.stats['hits'] == 2 and st.stats['decoded_here'] == 3
    st.request([names[5], names[6], names[1]])                          # ahead of time: only the two absent ones decode
    st.offer(names[7], torch.from_numpy(pix(7)))                        # a DataLoader worker's pixels
    st.offer(names[1], torch.from_numpy(pix(1)))                        # resident already: ignored
    check([5, 6, 7, 1])
    assert st.stats['decoded_here'] == 5 and st.stats['decoded_by_workers'] == 1
    # 8 slots hold 0,1,2,5,6,7: two more fit, the third replaces the OLDEST (0) but never an image of the step being assembled
    check([0, 8, 9, 10, 11])                                            # 0 is a hit and must survive although it is first in line
    assert st.stats['evicted'] == 2                                     # 1 and 2 went (0 was skipped)
    assert not bool(st.flags[1]) and not bool(st.flags[2]) and bool(st.flags[0])
    check([1, 2, 0, 11])                                                # evicted images come back by decoding again
    with pytest.raises(RuntimeError):
        st.batch(names[:9] + names[12:20])                              # more distinct new images than slots
    st.close()


def _file_trainer(tmp_path, tag, image_store, n_workers=0, batch_size=12, n_train=40, seed_imgs=0, **kw):
    from PIL import Image
    from test_host_cpu import _fake_loaders
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, n_train, 8)
    d = os.path.join(str(tmp_path), 'imgs'); os.makedirs(d, exist_ok=True)
    r = np.random.RandomState(seed_imgs)
    for split in dl.values():
        for b in split:
            paths = []
            for nm in b['image_filename']:
                p = os.path.join(d, nm + ('.png' if int(nm[4:]) % 2 else '.jpg'))
                if not os.path.exists(p):
                    j = int(nm[4:])
                    yy, xx = np.mgrid[0:48 + j % 5, 0:64 + j % 7]
                    img = np.stack([(xx * 3 + j * 11) % 256, (yy * 5 + j * 7) % 256, ((xx + yy) * 2 + j) % 256], axis=2) + r.randint(0, 20, size=yy.shape + (3,))
                    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(p)
                paths.append(p)
            b['path_to_image'] = paths
    gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 3, {}, 0.05, True, K=0.1, use_CNN=True)
    tr = oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=n_workers,
                              batch_size=batch_size, experiment_name=tag, embedding_dim=10, neg_to_pos_ratio=3, image_fc7=None,
                              normalize=None, alpha=0.05, experiment_dir=str(tmp_path), n_epochs=1, eval_interval=1, image_store=image_store, **kw)
    return tr, crit, gd, dl


def _capture_batches(tr):
    seen = []
    orig = tr.img_feat_net.forward_raw
    def spy(x, split=None):
        seen.append(x.detach().clone())
        return orig(x, split=split)
    tr.img_feat_net.forward_raw = spy
    return seen


def test_store_rows_are_bit_equal_to_get_image_and_getitem(tmp_path):
    tr, crit, gd, dl = _file_trainer(tmp_path, 'a', True)
    st = tr.image_store
    assert st is not None and st.capacity == 40 + 16
    ds = tr.datasets['train']
    names = [n for n in st.names[:9]]
    cold = st.batch(names)                                              # decoded by the pool, uploaded, gathered
    warm = st.batch(names)                                              # resident
    ahead_names = st.names[9:14]
    st.request(ahead_names); ahead = st.batch(ahead_names)              # requested ahead of use
    for got, nms in ((cold, names), (warm, names), (ahead, ahead_names)):
        want = torch.stack([ds.get_image(n) for n in nms])              # the reference's API: host float tensors (oe_h.py:668-677)
        assert got.shape == want.shape and torch.equal(got.cpu(), want)
        assert got.is_contiguous(memory_format=torch.channels_last)
    # c_out = 4: the stem's operand, zero 4th channel
    g4 = st.gather(st.resolve(names), None, c_out=4)
    assert torch.equal(g4[:, :3].cpu(), torch.stack([ds.get_image(n) for n in names])) and not g4[:, 3].any()
    # a train item: the ref's flip applied on the GPU == the tensor path's transform on the host
    items = [i for i in range(len(ds)) if type(ds.edge_of(i)[1]) == str][:10]
    torch.manual_seed(5); refs = [ds[i]['to'] for i in items]
    sv, ds.store_view = ds.store_view, None
    torch.manual_seed(5); tens = [ds[i]['to'] for i in items]
    ds.store_view = sv
    assert any(r.flip for r in refs)
    got = st.batch([r.name for r in refs], [r.flip for r in refs])
    assert torch.equal(got.cpu(), torch.stack(tens))


@pytest.mark.parametrize('lookahead', [True, False])
def test_trainer_from_files_store_path_equals_host_tensor_path(tmp_path, lookahead):
    """Three steps of JointEmbeddings.train_epoch from image files: with the HBM image store (ImageRef items, one gather per step,
    negatives' images decoded ahead) and with the reference's host tensors (image_store=False).  Same negatives (bit-exact stream), the CNN
    batch of every step bit-equal, losses equal to the run-to-run noise of the float atomics in the weight gradients."""
    runs = {}
    for tag, use_store in (('store', True), ('host', False)):
        torch.manual_seed(0)
        tr, crit, gd, dl = _file_trainer(tmp_path, tag, use_store)
        tr.negative_lookahead = lookahead
        seen = _capture_batches(tr)
        negs, losses = [], []
        orig_step = tr.train_step
        def step(item, _o=orig_step, _c=crit):
            out = _o(item)
            negs.append(_c.last_negatives.copy()); losses.append(out[0])
            return out
        tr.train_step = step
        torch.manual_seed(123)                                          # the flips of the epoch
        running, steps = tr.train_epoch(max_steps=3)
        torch.cuda.synchronize()
        assert steps == 3
        runs[tag] = (seen, negs, [float(l) for l in losses], float(running), tr)
    s, h = runs['store'], runs['host']
    assert runs['store'][4].image_store is not None and runs['host'][4].image_store is None
    for a, b in zip(s[1], h[1]):
        assert np.array_equal(a, b)
    assert len(s[0]) == 3 and len(h[0]) == 3
    for a, b in zip(s[0], h[0]):
        assert a.shape == b.shape and torch.equal(a, b.to(a.device))
    st = runs['store'][4].image_store.stats
    assert st['decoded_here'] > 0 and st['uploads'] >= 1
    assert s[2][0] == h[2][0]                                           # same weights, same batch, a forward without atomics: the same loss
    np.testing.assert_allclose(s[2], h[2], rtol=3e-2)                   # later steps: the weight gradients' float atomics, through Adam at lr 1e-3


def test_trainer_from_files_decodes_ahead_on_the_stores_threads(tmp_path):
    """With the image store the train loader forks NO worker processes (JointEmbeddings.persistent_workers' comment: the measured fork stall
    and deadlock); n_workers sizes the store's decode pool, and train_epoch's lookahead requests every file of step t+1 -- positives and
    negatives -- while step t runs.  Every CNN row of a step is the oracle's tensor of its file (mirrored or not for positives, never for
    negatives); a second pass over the same positives decodes nothing for them."""
    torch.manual_seed(0)
    tr, crit, gd, dl = _file_trainer(tmp_path, 'w', True, n_workers=3)
    assert tr.dataloaders['train'].num_workers == 0 and tr.image_store._pool._max_workers == 4
    seen = _capture_batches(tr)
    rows_of = []
    orig_batch = crit._image_batch
    def spy(rows, dev, store):
        rows_of.append(list(rows)); return orig_batch(rows, dev, store)
    crit._image_batch = spy
    running, steps = tr.train_epoch(max_steps=4)
    torch.cuda.synchronize()
    st = tr.image_store
    assert steps == 4 and st.stats['decoded_by_workers'] == 0 and st.stats['decoded_here'] > 0
    locs = st.locs
    n_pos = 12
    for rows, x in zip(rows_of, seen):
        assert all(isinstance(r, tuple) for r in rows)
        x = x.cpu().numpy()
        for j, (name, flip) in enumerate(rows):
            want = O.image_batch(decode_u8(locs[name], 224)[None], [flip])[0]
            assert np.array_equal(x[j], want)
            if j >= n_pos:
                assert not flip                                         # images drawn as negatives: get_image's transform (no flip)
    # second pass over the same images: everything is resident
    before = dict(st.stats)
    tr.epoch = 0
    tr.train_epoch(max_steps=4)
    torch.cuda.synchronize()
    assert st.stats['hits'] > before['hits'] and st.stats['decoded_here'] - before['decoded_here'] < 12      # (a new negative may still meet an image for the first time)
    assert np.isfinite(float(running))


def test_reference_exact_batches_from_files_through_the_store(tmp_path):
    """reference_exact_batches=True fed from image files: the reference's four forwards are assembled from the store (positives as ImageRef
    handles with their flip, every fixed image end K more times by name, unflipped), and one step runs."""
    torch.manual_seed(0)
    tr, crit, gd, dl = _file_trainer(tmp_path, 'x', True, reference_exact_batches=True)
    assert tr.reference_exact_batches and crit.reference_exact_batches and tr.cnn_passes == 1
    batches = []
    hk = tr.img_feat_net.model.conv1.register_forward_hook(lambda m_, i_, o_: batches.append(int(i_[0].shape[0])))
    rows_of = []
    orig_batch = crit._image_batch
    def spy(rows, dev, store):
        rows_of.append(list(rows)); return orig_batch(rows, dev, store)
    crit._image_batch = spy
    running, steps = tr.train_epoch(max_steps=1)
    hk.remove(); torch.cuda.synchronize()
    assert steps == 1 and np.isfinite(float(running))
    neg = crit.last_negatives
    N = tr.n_classes
    n_img_pos = sum(1 for r in rows_of[0] if isinstance(r, tuple))        # the positives' to-side forward comes first (no image on a from side here)
    assert batches[0] == n_img_pos and sum(batches) == crit.last_cnn_rows
    assert crit.last_cnn_rows == n_img_pos + int((neg >= N).sum()) + crit.neg_to_pos_ratio * n_img_pos
    assert all(isinstance(r, tuple) for rows in rows_of for r in rows)     # every row came from the store
    assert not any(flip for rows in rows_of[1:] for _, flip in rows)       # re-embedded fixed ends and image negatives: get_image's transform, never mirrored
