"""The reference's own call sites, run against this package on the GPU: criterion / Embedder / trainer objects with the
reference's names and signatures, checked against the reference-generated fixtures and the oracle."""
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN
from oracle import cone_oracle as O

pytestmark = pytest.mark.gpu

from learning_embeddings_amd import oe_h, embed_toy, order_embeddings, loss as loss_mod, experiment, ops  # noqa: E402
from learning_embeddings_amd.hierarchy import NegativeGraph, SyntheticLabelMap  # noqa: E402
from learning_embeddings_amd.engine import StepEngine  # noqa: E402

DEV = 'cuda'


class _IdentityCNN(torch.nn.Module):
    """Stand-in for FeatCNN18 exactly as in the fixture generator: the "image" IS its raw CNN output row; the criterion
    applies FeatCNN18.soft_clip (fused in the kernel)."""
    def __init__(self, K):
        super().__init__(); self.K = K
    def forward_raw(self, x):
        return x.float()
    def forward(self, x):
        return ops.ImageSoftClipFn.apply(x.float(), self.K)


@pytest.mark.parametrize('tag', ['s3', 'ethec'])
def test_criterion_call_site_vs_reference_fixture(tag):
    f = np.load(os.path.join(GOLDEN, 'F5_criterion.npz'))
    g = lambda k: f[tag + '_' + k]
    lm = SyntheticLabelMap(g('levels').tolist(), edges=[tuple(e) for e in g('edges').tolist()])
    N, M, Kn = lm.n_classes, int(g('n_images')), int(g('Kneg'))
    names = ['img_%06d' % j for j in range(M)]
    n2i = {i: i for i in range(N)}; n2i.update({names[j]: N + j for j in range(M)}); i2n = {v: k for k, v in n2i.items()}
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, Kn, {}, float(g('alpha')), bool(g('pick_per_level')), K=float(g('K')), use_CNN=True)
    crit.set_negative_graph(NegativeGraph.from_labelmap(lm, n_images=M, pick_per_level=bool(g('pick_per_level')), seed=0), n2i, i2n)
    R = torch.tensor(g('R'), device=DEV, requires_grad=True)

    class DL:
        def get_image(self, fname):
            return R[n2i[fname] - N]
    crit.set_dataloader(DL())
    model = oe_h.Embedder(g('W').shape[1], lm, None, K=float(g('K'))).to(DEV)
    with torch.no_grad():
        model.embeddings.weight.copy_(torch.tensor(g('W')))
    of = [i2n[int(i)] for i in g('from')]; ot = [i2n[int(i)] for i in g('to')]
    inputs_to = [R[n2i[t] - N] if isinstance(t, str) else t for t in ot]
    crit.seed_sampler(0)                                                    # random.seed(0) right before the call
    loss, e_pos, e_neg = crit(model, _IdentityCNN(float(g('K'))), list(of), inputs_to, of, ot, torch.ones(len(of)), 'train')
    assert np.array_equal(crit.last_negatives, g('neg'))                    # bit-exact negative selection
    assert e_neg.shape == g('e_neg').shape and e_pos.shape == g('e_pos').shape
    loss.backward()
    assert np.abs(e_pos.detach().cpu().numpy() - g('e_pos')).max() <= 1e-4
    assert np.abs(e_neg.detach().cpu().numpy() - g('e_neg')).max() <= 1e-4
    assert abs(loss.item() - float(g('loss'))) <= 1e-4 * abs(float(g('loss')))
    gW = model.embeddings.weight.grad.cpu().numpy()
    assert np.abs(gW - g('gW')).max() / np.abs(g('gW')).max() < 1e-3
    assert np.abs(R.grad.cpu().numpy() - g('gR')).max() / np.abs(g('gR')).max() < 1e-3
    # the reference's single-draw API, same stream
    crit.seed_sampler(0)
    assert crit.sample_negative_edge(u=of[0], v=None, level_id=0) == int(g('neg')[0, 0])


def test_criterion_eval_branch_and_E_operator():
    lm = SyntheticLabelMap([2, 4, 8])
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 2, {}, 0.05, False, K=0.1, use_CNN=True)
    model = oe_h.Embedder(10, lm, None, K=0.1).to(DEV)
    f = torch.rand(5, 5, 10, device=DEV) * 0.3; t = torch.rand(5, 5, 10, device=DEV) * 0.3
    loss, e_pos, e_neg = crit(model, None, f, t, None, None, None, 'val')   # pre-built [B, 1+2K, D] tensors (oe_h.py:908-925)
    fo, to = f.cpu().numpy(), t.cpu().numpy()
    want_pos = O.cone_energy(fo[:, 0], to[:, 0], 0.1); want_neg = O.cone_energy(fo[:, 1:], to[:, 1:], 0.1)
    assert np.abs(e_pos.cpu().numpy() - want_pos).max() < 1e-4 and np.abs(e_neg.cpu().numpy() - want_neg).max() < 1e-4
    assert abs(loss.item() - (want_pos.sum() + np.maximum(0.05 - want_neg, 0).sum())) < 1e-3
    emb = model(torch.tensor([[0, 3], [5, 7]], device=DEV))
    assert emb.shape == (2, 2, 10)
    assert np.abs(emb.detach().cpu().numpy().reshape(4, 10) - O.embedder_forward(model.embeddings.weight.detach().cpu().numpy(), [0, 3, 5, 7], 0.1)).max() < 2e-6


def test_config1_toy_order_embedding_step():
    f = np.load(os.path.join(GOLDEN, 'F7_order_embedding.npz'))
    g_ = embed_toy.ToyGraph(levels=4, branching_factor=3)
    crit = order_embeddings.OrderEmbeddingLoss(g_, neg_to_pos_ratio=4, alpha=1.0, pick_per_level=True)
    tr = embed_toy.ToyOrderEmbedding(g_, crit, lr=0.1, batch_size=16, embedding_dim=6, neg_to_pos_ratio=4)
    with torch.no_grad():
        tr.model.embeddings.weight.copy_(torch.tensor(f['toy3_W']))
    W0 = f['toy3_W'].copy()
    loss, e_pos, e_neg = tr.train_step(f['toy3_from'].tolist(), f['toy3_to'].tolist())
    assert np.array_equal(crit.last_negatives, f['toy3_neg'])
    assert abs(loss.item() - float(f['toy3_loss'])) < 1e-4 * abs(float(f['toy3_loss']))
    assert np.abs(e_neg.cpu().numpy() - f['toy3_e_neg']).max() < 1e-5
    # the update is a plain Adam step on the reference gradient (no Riemannian rescale, no clip in the Euclidean trainer)
    Wo, _, _ = O.adam_update(W0, f['toy3_gW'], np.zeros_like(W0), np.zeros_like(W0), 1, 0.1)
    assert np.abs(tr.model.embeddings.weight.detach().cpu().numpy() - Wo).max() < 1e-5


def test_config4_multilevel_ce_trainer_step():
    lm = SyntheticLabelMap.ethec()
    crit = loss_mod.MultiLevelCELoss(lm)
    exp = experiment.ETHECExperiment({}, lm, crit, lr=1e-3, batch_size=8, model_name='resnet18', experiment_dir='/tmp/lec_exp',
                                     compute_dtype=torch.float32)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(8, 3, 32, 32, generator=g)
    lvl = torch.stack([torch.randint(0, n, (8,), generator=g) for n in lm.levels], 1)
    w0 = exp.arena.data.clone()
    l1, out = exp.train_step(x, None, lvl)
    assert out.shape == (8, 723) and torch.isfinite(l1)
    ol, _ = O.multilevel_ce(out.detach().float().cpu().numpy(), lvl.numpy(), lm.levels)
    assert abs(l1.item() - ol) < 1e-4 * ol
    assert (exp.arena.data - w0).abs().max().item() > 0                     # parameters moved
    l2, _ = exp.train_step(x, None, lvl)
    assert l2.item() < l1.item()                                            # same batch again: the loss goes down


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_config4_full_size_step_resnet50_b512(dtype):
    """BASELINE.json configs[3] at its stated size: ResNet-50, 723 ETHEC logits, batch 512, 224 x 224, through
    engine.ClassifierEngine (the `bench.py --workload cfg4` path): the loss of every step equals the oracle's multi-level CE of
    the logits the step produced, the hipGraph replay continues the eager steps, and training on a fixed pool reduces the loss."""
    from learning_embeddings_amd.engine import ClassifierEngine
    eng = ClassifierEngine('cfg4', dtype=dtype, use_graph=True, graph_after=2)
    losses = []
    for s in range(5):
        l = eng.step()
        loss, out = eng.last
        assert out.shape == (512, 723)
        lvl = eng.pool_levels.index_select(0, eng.idx_dev).cpu().numpy()
        ol, _ = O.multilevel_ce(out.detach().float().cpu().numpy(), lvl, eng.labelmap.levels)
        assert abs(float(loss) - ol) <= 2e-4 * ol, (s, float(loss), ol)
        losses.append(float(l))
    assert eng.hip_graph is not None, eng.graph_error
    assert sum(eng.library_conv_launches_per_step.values()) == 0, eng.library_conv_launches_per_step
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_step_engine_matches_oracle_stream_and_loss():
    eng = StepEngine('tiny', n_images=64, dtype='fp32')
    lm = eng.labelmap
    leaf = [lm.level_start[-1] + (j % lm.levels[-1]) for j in range(64)]
    A = O.dense_negative_adjacency(lm.n_classes, sorted(lm.edges), leaf)
    smp = O.DenseSampler(A, lm.levels, pick_per_level=True, seed=0)
    m_prev = np.zeros_like(eng.table.cpu().numpy()); v_prev = m_prev.copy()
    for s in range(3):
        W0 = eng.table.cpu().numpy().copy()
        got = {}
        h = eng.img_feat_net.model.fc.register_forward_hook(lambda m, i, o: got.__setitem__('f', o.detach().float().cpu().numpy()))
        eng.step(); torch.cuda.synchronize(); h.remove()
        loss, e_pos, e_neg, frm, to, neg = eng.last
        want = smp.draw_batch(frm, to, eng.K)
        assert np.array_equal(neg, want)                                    # engine's threaded prefetch == reference stream
        B, N = eng.B, eng.N
        neg_o = neg.astype(np.int64).copy()
        cols = np.asarray(eng.img_passes)
        neg_o[:, cols] = N + B + np.arange(B)[:, None] * eng.cnt + np.arange(eng.cnt)[None, :]
        o = O.joint_loss_fwd_bwd(W0, got['f'], frm, N + np.arange(B), neg_o, eng.alpha, eng.K_cone)
        assert abs(loss.item() - o[0]) <= 1e-4 * max(1, abs(o[0]))
        assert np.abs(e_neg.cpu().numpy() - o[2]).max() <= 1e-4
        # table update = the oracle's rescale -> Adam -> clip on the oracle's gradient
        Wn, m_prev, v_prev = O.table_step_adam(W0, o[3].astype(np.float32), m_prev, v_prev, s + 1, eng.lr, eng.K_cone)
        assert np.abs(eng.table.cpu().numpy() - Wn).max() < 5e-6
    eng.close()


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_step_engine_hipgraph_replay_matches_eager_launches(dtype):
    """Launch mode only.  (1) On the SAME parameters and the SAME uploaded batch a replay of the captured graph (forward +
    loss + backward, two streams) and an eager run of the same region give the same loss, energies, table gradient and
    CNN gradient (to the run-to-run noise of bf16 atomics).  (2) Over a short training run the replays see every fresh
    index upload: same negatives as the eager engine, loss trajectory within the eager engine's own run-to-run drift."""
    runs = {}
    for mode in (False, True):
        torch.manual_seed(0)
        eng = StepEngine('tiny', n_images=64, dtype=dtype, use_graph=mode, graph_after=2)
        losses, negs = [], []
        for _ in range(6):
            eng.step(); losses.append(eng.last[0].clone()); negs.append(eng.last[5].copy())
        torch.cuda.synchronize()
        assert (eng.hip_graph is not None) == mode, eng.graph_error
        runs[mode] = (torch.stack(losses).flatten().cpu().numpy(), negs, eng.table.cpu().numpy().copy())
        if mode:                                                            # (1): same state, same batch, both launch modes
            eng.hip_graph.replay(); torch.cuda.synchronize()
            l1, ep1, en1 = [t.clone() for t in eng.graph_out]
            gt1, ga1 = eng.table_grad.clone(), eng.arena.grad.clone()
            l2, ep2, en2 = eng._core(None); torch.cuda.synchronize()
            # bf16 backbone at random init: MIOpen / hipBLASLt may pick other kernels for the captured launches than for
            # the eager ones, and a few ulps of bf16 move this tiny network's loss by ~1 %; a stale batch or a missed
            # kernel would move it by O(1)
            cosf = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()
            if dtype == 'fp32':
                # the reference's precision: every kernel of the step is liblecone's and the forward has no atomics -> the replay's
                # loss and energies EQUAL the eager ones; the gradients differ by the order of their float atomics only
                assert torch.equal(l1, l2) and torch.equal(ep1, ep2) and torch.equal(en1, en2)
                rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
                if rel(ga1, eng.arena.grad) >= 1e-4:                        # say WHICH parameters differ before failing
                    names = [n for n, p_ in eng.img_feat_net.named_parameters() if p_.requires_grad]
                    for k_, (n_, o_) in enumerate(zip(names, eng.arena.offsets)):
                        num = eng.arena.params[k_].numel()
                        r_ = rel(ga1[o_:o_ + num], eng.arena.grad[o_:o_ + num])
                        if r_ > 1e-4:
                            print('%-36s rel %.3g  |replay| %.4g  |eager| %.4g' % (n_, r_, ga1[o_:o_ + num].norm().item(), eng.arena.grad[o_:o_ + num].norm().item()))
                    print('pass streams', [s_.cuda_stream for s_ in eng.pass_streams], 'passes', eng.passes)
                assert rel(gt1, eng.table_grad) < 1e-5 and rel(ga1, eng.arena.grad) < 1e-4
            else:
                assert abs(l1.item() - l2.item()) <= 5e-2 * abs(l2.item())
                assert cosf(en1, en2) > 0.995 and cosf(ep1, ep2) > 0.995
                assert cosf(gt1, eng.table_grad) > 0.99
                assert cosf(ga1, eng.arena.grad) > 0.97
        eng.set_launch_mode(False)                                          # eager probe steps after replays (bench.py does this)
        eng.step(); torch.cuda.synchronize()
        assert torch.isfinite(eng.last[0]).all()
        eng.set_launch_mode(True); eng.step(); torch.cuda.synchronize()
        assert (eng.hip_graph is not None) == mode
        eng.close()
        assert eng.hip_graph is None and eng._graph_saved is None and eng.graph_out is None      # close() destroys the graphs, not the GC later
    le, lg = runs[False][0], runs[True][0]
    assert all(np.array_equal(a, b) for a, b in zip(runs[False][1], runs[True][1]))
    # two runs of the same training agree while their gradients are bit-equal; float-atomic summation order (weight gradients, and
    # at fp32 the two concurrent half-batch passes) is not reproducible, Adam's first steps move every weight by +-lr whatever the
    # size of its gradient, and this workload's BatchNorm batches are 8 rows of 1 x 1 pixels in layer4: the trajectories part after
    # two steps (bf16: two eager runs drift by ~5 % here on their own)
    assert np.abs(le - lg)[:2].max() <= (1e-5 if dtype == 'fp32' else 0.15) * np.abs(le).max()
    assert np.abs(le - lg).max() <= 0.15 * np.abs(le).max()
    assert len(set(np.round(lg, 4).tolist())) > 3                           # not a stale batch replayed over and over
    assert np.abs(runs[False][2] - runs[True][2]).max() < (5e-5 if dtype == 'fp32' else 5e-3)


def test_joint_embeddings_trainer_runs_and_learns(tmp_path):
    from test_host_cpu import _fake_loaders
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, 32, 8)
    for split in dl.values():                                               # real-looking images: 3x32x32 in [0,1)
        for b in split:
            b['path_to_image'] = [torch.rand(3, 32, 32, generator=torch.Generator().manual_seed(int(n[4:]))) for n in b['image_filename']]
    gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.05, True, K=0.1, use_CNN=True)
    tr = oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=0,
                              batch_size=16, experiment_name='t', embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                              normalize=None, alpha=0.05, experiment_dir=str(tmp_path), n_epochs=2, eval_interval=1)
    assert len(tr.datasets['train']) == gd['G_train_tc'].size()
    # train_step never synchronises by itself, but it lets the host run at most two steps ahead of the GPU (the allocator holds
    # what the side stream touched until its events pass: unbounded run-ahead piled up one step of activations per step)
    crit.set_dataloader(tr.datasets['train']); tr.model.train(); tr.img_feat_net.train()
    it = iter(tr.dataloaders['train'])
    for _ in range(5):
        tr.train_step(next(it))
        assert len(tr._steps_in_flight) <= 2
    torch.cuda.synchronize()
    assert all(e.query() for e in tr._steps_in_flight)
    tr.run_model()
    assert np.isfinite(tr.last_epoch_loss)
    n = tr.model.embeddings.weight.detach().norm(dim=1)
    assert n.min().item() >= O.inner_radius(0.1) - 1e-6 and n.max().item() <= 1.0      # oe_h.py:1771 clip invariant
    assert set(['m-f1', 'hit@1', 'hit@5', 'M-f1']).issubset(tr.last_metrics)
    assert os.path.exists(os.path.join(tr.path_to_save_model, 'best_model_model.pth'))
    tr.load_model('best_model')
    sd = torch.load(os.path.join(tr.path_to_save_model, '0_model.pth'))['model_state_dict']
    assert list(sd) == ['module.embeddings.weight']                          # the reference's DataParallel key prefix


def test_embed_images_replayed_forward_equals_eager_forward(tmp_path):
    """JointEmbeddings.embed_images replays the forward of a full chunk as a hipGraph from its third occurrence on (the reference's 'train'-phase
    chunks are 10 images: launch-bound).  Against eval_graphs = False on a copy of the same networks: eval mode -- rows bit-equal, also after the
    weights and running statistics have moved (the captured graph holds the BatchNorm layers' scale / shift vectors, refreshed in place);
    train mode -- rows bit-equal chunk by chunk and the running statistics after all chunks equal (every replay updates them like a forward)."""
    import copy
    from test_host_cpu import _fake_loaders
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, 48, 8)
    for split in dl.values():
        for b in split:
            b['path_to_image'] = [torch.rand(3, 32, 32, generator=torch.Generator().manual_seed(int(n[4:]))) for n in b['image_filename']]
    gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.05, True, K=0.1, use_CNN=True)
    tr = oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=0,
                              batch_size=16, experiment_name='t', embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                              normalize=None, alpha=0.05, experiment_dir=str(tmp_path), n_epochs=1, eval_interval=1)
    crit.set_dataloader(tr.datasets['train'])
    names = [n for n in gd['G_train'] if type(n) == str]
    assert len(names) >= 44
    names = names[:44]                                                       # 5 full chunks of 8 (two eager, capture + replay, two replays) and one of 4 (eager)
    with torch.no_grad():
        for m in tr.img_feat_net.modules():
            if hasattr(m, 'eval_affine'):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    for train in (False, True):
        tr.img_feat_net.train(train)
        ref_net = copy.deepcopy(tr.img_feat_net)
        tr.drop_eval_graphs()
        with torch.no_grad():
            got = tr.embed_images(names, bs=8)
            assert tr._eval_graphs[((8, 3, 32, 32), train)]['graph'] is not None
            live, tr.img_feat_net, tr.eval_graphs = tr.img_feat_net, ref_net, False
            try:
                want = tr.embed_images(names, bs=8)
            finally:
                tr.img_feat_net, tr.eval_graphs = live, True
        assert torch.equal(got, want), 'train=%s' % train
        for (k, a), (_, b) in zip(live.named_buffers(), ref_net.named_buffers()):
            assert torch.equal(a, b), k
        if not train:
            # parameters and running statistics move (a training epoch between two evaluation phases): the SAME graph must follow
            with torch.no_grad():
                for net in (live, ref_net):
                    g = torch.Generator(device='cpu').manual_seed(5)
                    for m in net.modules():
                        if hasattr(m, 'eval_affine'):
                            m.running_var.mul_(1.5); m.running_mean.add_(0.05); m.weight.mul_(0.9)
                    for prm in net.parameters():
                        prm.add_(torch.randn(prm.shape, generator=g).to(prm.device) * 1e-3)
                gid = id(tr._eval_graphs[((8, 3, 32, 32), False)]['graph'])
                got2 = tr.embed_images(names, bs=8)
                assert id(tr._eval_graphs[((8, 3, 32, 32), False)]['graph']) == gid
                tr.img_feat_net, tr.eval_graphs = ref_net, False
                try:
                    want2 = tr.embed_images(names, bs=8)
                finally:
                    tr.img_feat_net, tr.eval_graphs = live, True
            assert torch.equal(got2, want2) and not torch.equal(got2, got)
            # ... and after REAL training steps: liblecone's optimizer and BatchNorm kernels write parameters and running statistics through raw
            # pointers (no tensor version counter moves), the replayed graph recomputes its scale / shift vectors from them on the stream
            crit.set_dataloader(tr.datasets['train']); tr.model.train(); tr.img_feat_net.train()
            it = iter(tr.dataloaders['train'])
            for _ in range(2):
                tr.train_step(next(it))
            tr.img_feat_net.eval()
            with torch.no_grad():
                got3 = tr.embed_images(names, bs=8)
                assert id(tr._eval_graphs[((8, 3, 32, 32), False)]['graph']) == gid
                tr.eval_graphs = False                              # (eval mode mutates nothing: the same network, launched eagerly, is the reference)
                try:
                    want3 = tr.embed_images(names, bs=8)
                finally:
                    tr.eval_graphs = True
            assert torch.equal(got3, want3) and not torch.equal(got3, got2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_joint_embeddings_fast_path_matches_plain_autograd(tmp_path, dtype):
    """JointEmbeddings.train_step through liblecone's convolutions / fused BatchNorm / arena gradients / side stream (the path
    bench.py measures) against the same trainer on plain autograd + library convolutions: same seed, same batches -> same loss,
    same label-table update, same image-network update (to the noise of the precision in use)."""
    from test_host_cpu import _fake_loaders
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, 32, 8)
    for split in dl.values():
        for b in split:
            b['path_to_image'] = [torch.rand(3, 64, 64, generator=torch.Generator().manual_seed(int(n[4:]))).to(DEV) for n in b['image_filename']]
    gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)
    out = {}
    for fast in (True, False):
        crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.05, True, K=0.1, use_CNN=True)
        tr = oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=0,
                                  batch_size=16, experiment_name='f%d' % fast, embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                                  normalize=None, alpha=0.05, experiment_dir=str(tmp_path), n_epochs=1, eval_interval=5,
                                  compute_dtype=dtype, fast_path=fast, cnn_passes=1)    # one CNN pass: the same BatchNorm batch on both sides
        assert (tr.overlap is not None) == fast
        crit.set_dataloader(tr.datasets['train'])
        tr.train_set.transform = None                                        # no random flip: identical batches on both sides
        tr.model.train(); tr.img_feat_net.train()
        signs = {}
        from learning_embeddings_amd.resnet import BatchNormAct2d
        for name, m in tr.img_feat_net.named_modules():
            if isinstance(m, BatchNormAct2d) and m.fuse_relu:
                m.register_forward_hook(lambda mod, inp, o, name=name: signs.__setitem__(name, ((o[0] if isinstance(o, tuple) else o).detach() > 0)))
        torch.manual_seed(1)
        it = iter(tr.dataloaders['train'])
        losses = [float(tr.train_step(next(it))[0])]
        torch.cuda.synchronize()
        # the image network is compared through its GRADIENT (what the two paths compute); Adam's first update is lr * sign(g),
        # which turns rounding noise on near-zero gradients into +-lr on the parameter
        out[fast] = (losses, tr.model.embeddings.weight.detach().clone(), tr.arena.grad.clone(), signs)
    tol = 2e-4 if dtype == torch.float32 else 5e-2
    for a, b in zip(out[True][0], out[False][0]):
        assert abs(a - b) <= tol * max(1.0, abs(b)), (out[True][0], out[False][0])
    assert (out[True][1] - out[False][1]).abs().max().item() <= (1e-5 if dtype == torch.float32 else 3e-3)
    d = (out[True][2] - out[False][2]).double().norm().item() / out[False][2].double().norm().item()
    # bf16: two different kernel sets on a random-init network differ by bf16 rounding noise amplified through 18 layers
    # (test_resnet_fused_path_matches_unfused_paths measures cosine 0.93-0.94 between ANY two bf16 paths); fp32: tight -- unless the two
    # forwards (1e-6 apart: different summation order) land on different sides of a ReLU somewhere: ONE activation of 3.6e-7 against 0.0 in
    # layer2.1.bn1 moved this 22-image gradient by 5.5e-3 when the small convolutions began to be cut along K.  The derivative is discontinuous
    # there, both gradients are right; the tight bound holds when every ReLU decision agrees.
    flips = sum(int((out[True][3][k] != out[False][3][k]).sum().item()) for k in out[True][3])
    assert dtype != torch.float32 or flips <= 4, flips
    assert d < ((1e-4 if flips == 0 else 3e-2) if dtype == torch.float32 else 0.5), (d, flips)


# ------------------------------------------------------------------------------------------------ DP on one GPU (gloo)
def _dp_trainer_worker(rank, world, port, tmp, q, half_half=False):
    import os, sys
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), LEC_DIST_BACKEND='gloo')
    from conftest import ROOT
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch as t
    from test_host_cpu import _fake_loaders
    from learning_embeddings_amd import oe_h as m
    from learning_embeddings_amd.hierarchy import SyntheticLabelMap as LM
    lm = LM([2, 4, 8])
    dl = _fake_loaders(lm, 32, 8)
    for split in dl.values():
        for b in split:
            b['path_to_image'] = [t.rand(3, 32, 32, generator=t.Generator().manual_seed(int(n[4:]))) for n in b['image_filename']]
    gd = m.create_combined_graphs(dl, lm, pick_per_level=True)
    crit = m.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.05, True, K=0.1, use_CNN=True)
    tr = m.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=0,
                           batch_size=8, experiment_name='dp', embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                           normalize=None, alpha=0.05, experiment_dir=os.path.join(tmp, 'r%d' % rank), n_epochs=1, eval_interval=5,
                           half_half=half_half)
    negs = []
    orig = crit.negative_G.draw_batch
    def spy(f, t_, k):
        out = orig(f, t_, k); negs.append(out.copy()); return out
    crit.negative_G.draw_batch = spy
    tr.pass_samples('train')
    q.put((rank, tr.model.embeddings.weight.detach().cpu().numpy(), tr.arena.data[:4096].cpu().numpy(), negs[:3], tr.last_epoch_loss))
    t.distributed.barrier(); t.distributed.destroy_process_group()


@pytest.mark.parametrize('half_half', [False, True])
def test_joint_embeddings_data_parallel_two_ranks_share_one_gpu(tmp_path, half_half):
    """(half_half=True: oe_h.py:683-726's alternating label-label / label-image batches under data parallelism -- every rank maps the
    global batch's item indices to edges with the dataset's own item -> edge rule.)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    procs = [ctx.Process(target=_dp_trainer_worker, args=(r, 2, port, str(tmp_path), q, half_half)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda r: r[0])
    for p in procs: p.join(120)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])     # replicas stay identical
    assert np.isfinite(res[0][4]) and res[0][4] == res[1][4]
    # both ranks walked the SAME global negative stream (global batch of 16 edges), i.e. the single-process stream
    for a, b in zip(res[0][3], res[1][3]):
        assert a.shape[0] == 16 and np.array_equal(a, b)


# ------------------------------------------------------------------------------------------------ BASELINE.json configs as parity cases
def test_config2_ethec_resnet18_step_matches_oracle():
    """configs[1]: real ETHEC label DAG (723 nodes, 4 levels), resnet18, hyperbolic cone loss -- reduced batch, full 224x224."""
    eng = StepEngine('cfg2', n_images=512, dtype='bf16', batch=8)
    lm = eng.labelmap
    assert lm.levels == [6, 21, 135, 561] and eng.cnt == 1 and eng.n_rows == 16
    leaf = [lm.level_start[-1] + (j % lm.levels[-1]) for j in range(512)]
    A = O.dense_negative_adjacency(lm.n_classes, sorted(lm.edges), leaf)
    smp = O.DenseSampler(A, lm.levels, pick_per_level=True, seed=0)
    for s in range(2):
        W0 = eng.table.cpu().numpy().copy(); got = {}
        h = eng.img_feat_net.model.fc.register_forward_hook(lambda m, i, o: got.__setitem__('f', o.detach().float().cpu().numpy()))
        eng.step(); torch.cuda.synchronize(); h.remove()
        loss, e_pos, e_neg, frm, to, neg = eng.last
        assert np.array_equal(neg, smp.draw_batch(frm, to, eng.K))            # bit-exact negatives on the real DAG
        B, N = eng.B, eng.N
        neg_o = neg.astype(np.int64).copy(); cols = np.asarray(eng.img_passes)
        neg_o[:, cols] = N + B + np.arange(B)[:, None] * eng.cnt + np.arange(eng.cnt)[None, :]
        o = O.joint_loss_fwd_bwd(W0, got['f'], frm, N + np.arange(B), neg_o, eng.alpha, eng.K_cone)
        assert abs(loss.item() - o[0]) <= 1e-4 * max(1, abs(o[0]))
        assert np.abs(e_pos.cpu().numpy() - o[1]).max() <= 1e-4 and np.abs(e_neg.cpu().numpy() - o[2]).max() <= 1e-4
    eng.close()


def _engine_vs_oracle(eng, n_images, steps=2, check_table=True):
    """Run `steps` engine steps; per step: negatives bit-equal to the pinned dense-matrix sampler's stream, loss / E+ / E- against
    the oracle on the raw CNN outputs the fused kernel consumed, the gradient that flows INTO the CNN (d loss / d raw outputs: what
    fc and the backbone back-propagate) against the oracle's, label-table update against the oracle's rescale -> Adam -> clip."""
    lm = eng.labelmap
    leaf = (lm.level_start[-1] + eng.img_leaf).tolist()
    A = O.dense_negative_adjacency(lm.n_classes, sorted(lm.edges), leaf)
    smp = O.DenseSampler(A, lm.levels, pick_per_level=True, seed=0)
    m_prev = np.zeros_like(eng.table.cpu().numpy()); v_prev = m_prev.copy()
    for s in range(steps):
        W0 = eng.table.cpu().numpy().copy()
        eng.step(); torch.cuda.synchronize()
        assert sum(eng.library_conv_launches_per_step.values()) == 0, ('a convolution of this step went to a library', eng.library_conv_launches_per_step)
        got = {'f': eng.last_feats.float().cpu().numpy()}                   # the raw CNN outputs of THIS step (static buffer under replay)
        loss, e_pos, e_neg, frm, to, neg = eng.last
        assert np.array_equal(neg, smp.draw_batch(frm, to, eng.K)), 'negatives differ from the reference stream at step %d' % s
        B, N = eng.B, eng.N
        neg_o = neg.astype(np.int64).copy(); cols = np.asarray(eng.img_passes)
        neg_o[:, cols] = N + B + np.arange(B)[:, None] * eng.cnt + np.arange(eng.cnt)[None, :]
        o = O.joint_loss_fwd_bwd(W0, got['f'], frm, N + np.arange(B), neg_o, eng.alpha, eng.K_cone)
        assert abs(loss.item() - o[0]) <= 1e-4 * max(1, abs(o[0]))
        assert np.abs(e_pos.cpu().numpy() - o[1]).max() <= 1e-4 and np.abs(e_neg.cpu().numpy() - o[2]).max() <= 1e-4
        gfeat = eng.gfeat[:o[4].shape[0]].cpu().numpy()                       # what feats.backward() was fed this step
        assert np.abs(gfeat - o[4]).max() <= 2e-3 * np.abs(o[4]).max(), 'gradient into the CNN differs from the oracle at step %d' % s
        if check_table:
            Wn, m_prev, v_prev = O.table_step_adam(W0, o[3].astype(np.float32), m_prev, v_prev, s + 1, eng.lr, eng.K_cone)
            assert np.abs(eng.table.cpu().numpy() - Wn).max() < 5e-6


def test_config2_full_batch_128_matches_oracle():
    """configs[1] at its stated size: real ETHEC DAG, ResNet-18, hyperbolic cone loss, B = 128, 224 x 224, fp32 (the reference's
    precision)."""
    eng = StepEngine('cfg2', n_images=1024, dtype='fp32')
    assert eng.B == 128 and eng.n_rows == 256
    _engine_vs_oracle(eng, 1024)
    eng.close()


@pytest.mark.parametrize('dtype,graph', [('fp32', False), ('fp32', True), ('bf16', True), ('fp32-x3', True)])
def test_config3_benchmarked_workload_full_batch_256_matches_oracle(dtype, graph, monkeypatch):
    """The workload bench.py measures (configs[2]: S3 hierarchy of 2 000 labels, ResNet-50, B = 256, K = 5, D = 10, 512 CNN rows per
    step), at full size, in the launch modes it is measured in: negatives bit-equal to the reference's stream, loss / energies /
    table update against the oracle at the embedding boundary.  (fp32 eager, fp32 hipGraph replay, bf16 hipGraph replay, fp32 with the
    split convolutions under hipGraph replay.)"""
    if dtype == 'fp32-x3':
        from learning_embeddings_amd import resnet as R
        monkeypatch.setattr(R, 'F32_MODE', 'x3'); dtype = 'fp32'
    eng = StepEngine('cfg3', n_images=4096, dtype=dtype, use_graph=graph, graph_after=1)
    assert eng.B == 256 and eng.n_rows == 512 and eng.N == 2000
    _engine_vs_oracle(eng, 4096, steps=3)
    assert (eng.hip_graph is not None) == graph, eng.graph_error
    eng.close()


def test_config3_over_the_ethec_label_dag_full_batch_256_matches_oracle():
    """BASELINE.json's configs[2] names ETHEC; SURVEY.md 8(d) gives the config its own 2 000-node hierarchy S3 (the bench line).  The same
    ResNet-50 / B = 256 / K = 5 step over the real ETHEC DAG (723 labels, fixture F9), fp32, eager launches (the mode the bench times):
    negatives bit-equal to the reference's stream, loss / energies / table update against the oracle."""
    eng = StepEngine('cfg3_ethec', n_images=4096, dtype='fp32', use_graph=False)
    assert eng.B == 256 and eng.n_rows == 512 and eng.N == 723 and eng.arch == 'resnet50'
    _engine_vs_oracle(eng, 4096, steps=2)
    eng.close()


def test_chunked_cnn_rows_step_equals_unchunked_step_up_to_batchnorm_batches():
    """engine cnn_chunk: the step's CNN rows pushed through the backbone a chunk at a time (forward without saved activations, the
    loss on all outputs, re-forward + backward per chunk).  With ONE chunk covering every row the result must equal the plain step
    (same BatchNorm batch): loss, label-table update and image-network gradients."""
    a = StepEngine('tiny', n_images=64, dtype='fp32', passes=1)
    b = StepEngine('tiny', n_images=64, dtype='fp32', cnn_chunk=10 ** 6)           # >= n_rows: normalised to "no chunking"
    assert b.cnn_chunk is None
    b.close()
    b = StepEngine('tiny', n_images=64, dtype='fp32', passes=1)
    b.cnn_chunk = b.n_rows                                                          # force the chunked code path with a single chunk
    b.overlap.accumulate = True
    for _ in range(2):
        la = a.step(); lb = b.step()
    torch.cuda.synchronize()
    assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(la)))
    assert (a.table - b.table).abs().max().item() < 1e-6
    d = (a.arena.grad - b.arena.grad).double().norm().item() / a.arena.grad.double().norm().item()
    assert d < 1e-4, d
    a.close(); b.close()


@pytest.mark.parametrize('dtype', ['bf16'])
def test_config5_as_stated_b256_k256_fp16_table_chunked_cnn(dtype):
    """BASELINE.json configs[4] at its stated size on ONE GPU: 50 000-node 8-level hierarchy, 256 negatives per positive, ResNet-50,
    B = 256, "fp16+fp32-master" (the label table read from its fp16 shadow, bf16 conv stack with fp32 master weights).  K = 256 over
    8 levels draws 28 image negatives per positive: 7 424 CNN rows per step, pushed through the backbone in chunks (cnn_chunk).
    Negatives of the WHOLE first batch (256 x 512) bit-equal to the reference's own stream over the real dense 54 096^2 matrix of this DAG (fixture
    F4b, tests/golden/make_golden_sampler_s5.py), the second step's against the oracle's matrix-free sampler walking the same MT19937 stream on;
    loss / energies against the oracle evaluated on the fp16-rounded table at the embedding boundary."""
    eng = StepEngine('cfg5', n_images=4096, dtype=dtype, table_dtype='fp16', use_graph=True, graph_after=1)      # step 0 launches its chunks eagerly, step 1 replays the chunk graph
    assert eng.B == 256 and eng.K == 256 and eng.N == 50000 and eng.cnt == 28 and eng.n_rows == 256 * 29
    assert eng.cnn_chunk is not None and eng.table_h is not None
    W16 = eng.table_h.float().cpu().numpy().copy()
    eng.step(); torch.cuda.synchronize()
    assert sum(eng.library_conv_launches_per_step.values()) == 0, ('config 5: a convolution went to a library', eng.library_conv_launches_per_step)
    loss, e_pos, e_neg, frm, to, neg = eng.last
    B, N = eng.B, eng.N
    cols = np.asarray(eng.img_passes)
    assert (neg[:, cols] >= N).all() and (np.delete(neg, cols, axis=1)[:, :eng.K - len(cols)] < N).all()
    z = np.load(os.path.join(GOLDEN, 'F4b_sampler_s5_step0.npz'))
    assert np.array_equal(eng.img_leaf, z['image_leaf']) and np.array_equal(frm, z['pos_from']) and np.array_equal(to, z['pos_to'])
    assert np.array_equal(neg, z['neg']), 'config 5: negatives of the first batch differ from the reference stream'
    lazy = O.LazyDenseSampler(eng.labelmap.levels, sorted(eng.labelmap.edges), eng.labelmap.level_start[-1] + eng.img_leaf, pick_per_level=True)
    lazy.rng.seed(0)
    assert np.array_equal(lazy.draw_batch(frm, to, eng.K), neg)                   # the oracle walks step 0 (and stands where step 1 starts)
    neg_o = neg.astype(np.int64).copy()
    neg_o[:, cols] = N + B + np.arange(B)[:, None] * eng.cnt + np.arange(eng.cnt)[None, :]
    o = O.joint_loss_fwd_bwd(W16, eng.last_feats.float().cpu().numpy(), frm, N + np.arange(B), neg_o, eng.alpha, eng.K_cone)
    assert abs(loss.item() - o[0]) <= 1e-4 * max(1, abs(o[0]))
    assert np.abs(e_pos.cpu().numpy() - o[1]).max() <= 1e-4 and np.abs(e_neg.cpu().numpy() - o[2]).max() <= 1e-4
    assert torch.equal(eng.table_h, eng.table.to(torch.float16))                  # the shadow follows the updated master
    l2 = eng.step(); torch.cuda.synchronize()
    assert torch.isfinite(l2) and eng.chunk_graph is not None, eng.graph_error
    frm1, to1, neg1 = eng.last[3:]
    nb = 64
    assert np.array_equal(lazy.draw_batch(frm1[:nb], to1[:nb], eng.K), neg1[:nb]), 'config 5: negatives of the second batch differ from the oracle stream'
    eng.close()


def test_chunked_step_replayed_as_one_graph_per_chunk_equals_eager_chunks():
    """The chunked step's launch mode: ONE hipGraph of (gather, forward, windowed loss, backward) captured once and replayed for every chunk of every step --
    the loss kernel reads its row window from device memory (lec_joint_loss_fwd_bwd_window's window_dev).  Same losses, energies and label table as the
    eagerly launched chunks; the image network's gradients equal to float-atomic order."""
    a = StepEngine('tiny', n_images=64, dtype='fp32', cnn_chunk=8, use_graph=False)
    b = StepEngine('tiny', n_images=64, dtype='fp32', cnn_chunk=8, use_graph=True, graph_after=1)
    assert a.cnn_chunk == 8 and b.cnn_chunk == 8 and a.n_rows_pad // 8 >= 2
    for s_ in range(4):
        la = a.step(); lb = b.step()
        torch.cuda.synchronize()
        assert (b.chunk_graph is not None) == (s_ >= 1), b.graph_error
        assert abs(float(la) - float(lb)) <= 1e-4 * max(1.0, abs(float(la))), (s_, float(la), float(lb))
        assert np.array_equal(a.last[5], b.last[5])
        assert (a.last[1] - b.last[1]).abs().max().item() <= 1e-4 and (a.last[2] - b.last[2]).abs().max().item() <= 1e-4     # (two trajectories: float-atomic noise through Adam)
    assert (a.table - b.table).abs().max().item() < 1e-4
    # (after four Adam steps the two trajectories differ by float-atomic summation noise that Adam's first steps amplify on weights whose gradient is
    # noise: the gradients agree in direction and to a few percent in norm, not to rounding)
    ga, gb = a.arena.grad.double(), b.arena.grad.double()
    assert float(ga @ gb / (ga.norm() * gb.norm())) > 0.99 and (ga - gb).norm().item() / ga.norm().item() < 0.15
    a.close(); b.close()


def test_config5_deep_hierarchy_256_negatives_loss_path():
    """configs[4]: 50k-node 8-level hierarchy, 256 negatives per positive: sampler + fused loss at the full K (the CNN is
    not part of this case: 29 images per positive do not fit one pass at B=256; the label-embedding path is what it stresses)."""
    from learning_embeddings_amd.hierarchy import SYNTHETIC
    lm = SyntheticLabelMap(SYNTHETIC['S5'])
    N, M, B, K, D = lm.n_classes, 4096, 24, 256, 10
    assert N == 50000 and len(lm.levels) == 8
    # spread the images over the whole leaf level (with j mod n_leaf all 4096 images would sit under one level-1 node,
    # whose slot-L candidate list would be empty -- python's random.choice raises there, and so does the sampler)
    leaf_of = lm.level_start[-1] + (np.arange(M, dtype=np.int64) * 9973) % lm.levels[-1]
    g = NegativeGraph.from_labelmap(lm, image_leaf=leaf_of, pick_per_level=True, seed=0)
    par = lm.parents()
    rs = np.random.RandomState(0)
    img = rs.randint(0, M, B); frm = []
    for b in range(B):
        v = int(leaf_of[img[b]])
        for _ in range(b % 8):
            v = par[v][0]
        frm.append(v)
    frm = np.array(frm, dtype=np.int32); to = (N + img).astype(np.int32)
    neg = g.draw_batch(frm, to, K)
    L = len(lm.levels)
    # window / non-membership properties of every draw (the dense oracle matrix would be 2.9 GB here)
    for p in range(K):
        slot = p % (L + 1)
        col_u, col_v = neg[:, p], neg[:, K + p]
        if slot < L:
            assert ((col_u >= lm.level_start[slot]) & (col_u < lm.level_stop[slot])).all()
            assert ((col_v >= lm.level_start[slot]) & (col_v < lm.level_stop[slot])).all()
        else:
            assert (col_u >= N).all() and (col_v < N).all()                  # label `u` -> images only; image `v` -> labels only
    assert (neg[:, :K] != to[:, None]).all() and (neg[:, K:] != frm[:, None]).all()
    torch.manual_seed(0)
    W = oe_h.Embedder(D, lm, None, K=0.1).embeddings.weight.detach().numpy().copy()
    R = (rs.randn(M, D) * 0.3).astype(np.float32)
    Wt = torch.tensor(W, device=DEV, requires_grad=True); Rt = torch.tensor(R, device=DEV, requires_grad=True)
    code = lambda a: torch.tensor(np.where(a < N, a, -1 - (a - N)), dtype=torch.int32, device=DEV)
    loss, e_pos, e_neg = ops.JointLossFn.apply(Wt, Rt, code(frm.astype(np.int64)), code(to.astype(np.int64)),
                                               code(neg.astype(np.int64)).contiguous(), None, 0.1, 0.01, 0, 1, 1)
    loss.backward()
    o = O.joint_loss_fwd_bwd(W, R, frm, to, neg, 0.01, 0.1)
    assert np.abs(e_neg.cpu().numpy() - o[2]).max() <= 1e-4 and abs(loss.item() - o[0]) <= 1e-4 * abs(o[0])
    assert np.abs(Wt.grad.cpu().numpy() - o[3]).max() / (np.abs(o[3]).max() + 1e-12) < 2e-3
    assert np.abs(Rt.grad.cpu().numpy() - o[4]).max() / (np.abs(o[4]).max() + 1e-12) < 2e-3


def test_checkpoint_interchange_with_torch_adam(tmp_path):
    """The reference saves/loads `torch.optim.Adam` state (oe_h.py:1876-1957).  A trainer restored from a checkpoint whose
    optimizer state was produced by torch.optim.Adam continues exactly like torch.optim.Adam would, and our own
    checkpoints load into torch.optim.Adam."""
    from test_host_cpu import _fake_loaders
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, 16, 4)
    for split in dl.values():
        for b in split:
            b['path_to_image'] = [torch.rand(3, 32, 32, generator=torch.Generator().manual_seed(int(n[4:]))) for n in b['image_filename']]
    gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)
    def make(name):
        crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.05, True, K=0.1, use_CNN=True)
        return oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=0,
                                    batch_size=16, experiment_name=name, embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                                    normalize=None, alpha=0.05, experiment_dir=str(tmp_path), n_epochs=1, eval_interval=5)
    tr = make('a')
    # a torch.optim.Adam over [table] + cnn params, stepped twice on random gradients = "reference-trained" state
    params = [tr.model.embeddings.weight] + list(tr.img_feat_net.parameters())
    ref = [p.detach().clone().requires_grad_(True) for p in params]
    opt = torch.optim.Adam([{'params': ref}], lr=1e-3)
    g = torch.Generator(device='cpu').manual_seed(0)
    grads = [[torch.randn(p.shape, generator=g).to(DEV) for p in ref] for _ in range(3)]
    for s in range(2):
        for p, gr in zip(ref, grads[s]): p.grad = gr.clone()
        opt.step()
    torch.save({'epoch': 0, 'model_state_dict': {'module.embeddings.weight': ref[0].detach()}, 'optimizer_state_dict': opt.state_dict(),
                'loss': 0.0, 'optimal_threshold': 0.0, 'reconstruction_scores': {'f1': 0, 'precision': 0, 'recall': 0, 'accuracy': 0, 'threshold': 0}},
               os.path.join(tr.path_to_save_model, '7_model.pth'))
    sd = {k: v for k, v in tr.img_feat_net.state_dict().items()}
    names = [n for n, _ in tr.img_feat_net.named_parameters()]
    for n, p in zip(names, ref[1:]): sd[n] = p.detach()
    # the reference's key names: its FeatCNN18 keeps the ResNet inside nn.DataParallel (oe_h.py:301) -> `model.module.*`
    sd = {k.replace('model.', 'model.module.', 1): v for k, v in sd.items()}
    assert 'model.module.conv1.weight' in sd
    opt_img = torch.optim.Adam([p.detach().clone().requires_grad_(True) for p in tr.img_feat_net.parameters()], lr=1e-3)
    torch.save({'epoch': 0, 'model_state_dict': sd, 'optimizer_state_dict': opt_img.state_dict(), 'loss': 0.0, 'optimal_threshold': 0.0,
                'reconstruction_scores': {}}, os.path.join(tr.path_to_save_model, '7_img_feat_net.pth'))
    tr.load_model(7)
    assert tr.table_step == 2 and tr.arena.step == 2
    # third step on both sides with the same gradients (plain Adam on the table: no Riemannian rescale / clip here)
    for p, gr in zip(ref, grads[2]): p.grad = gr.clone()
    opt.step()
    tr.table_grad.copy_(grads[2][0])
    for p, gr in zip(tr.arena.params, grads[2][1:]): p.grad.copy_(gr)
    tr.table_step += 1
    ops.table_step_adam(tr.model.embeddings.weight.data, tr.table_grad, tr.table_m, tr.table_v, tr.table_step, 1e-3, 0.0, riemannian=False, clip=False)
    tr.arena.adam_step(1e-3)
    assert (tr.model.embeddings.weight.data - ref[0].detach()).abs().max().item() < 1e-6
    for p, r in zip(tr.arena.params, ref[1:]):
        assert (p.data - r.detach()).abs().max().item() < 1e-6
    # and back: our checkpoint's optimizer state loads into torch.optim.Adam
    tr.save_model(0.0, filename='out')
    ck = torch.load(os.path.join(tr.path_to_save_model, 'out_model.pth'))
    opt2 = torch.optim.Adam([{'params': [p.detach().clone().requires_grad_(True) for p in params]}], lr=1e-3)
    opt2.load_state_dict(ck['optimizer_state_dict'])
    assert int(opt2.state_dict()['state'][0]['step']) == 3
    # the image network's file: the reference's key names, and an optimizer entry its load_model can hand to
    # optimizer_images.load_state_dict (oe_h.py:1955-1956)
    ck2 = torch.load(os.path.join(tr.path_to_save_model, 'out_img_feat_net.pth'))
    assert all(k.startswith('model.module.') for k in ck2['model_state_dict'])
    import torch.nn as nn
    from learning_embeddings_amd.resnet import resnet18
    class RefShape(nn.Module):                     # the reference's module nesting: FeatCNN18.model = nn.DataParallel(resnet18) with fc -> D
        def __init__(self):
            super().__init__()
            m = resnet18(); m.fc = nn.Linear(512, 10); self.model = nn.DataParallel(m)
    ref_net = RefShape()
    ref_net.load_state_dict(ck2['model_state_dict'])                       # strict: every key present, none unexpected
    opt3 = torch.optim.Adam(ref_net.parameters(), lr=1e-3)
    opt3.load_state_dict(ck2['optimizer_state_dict'])


def test_order_embeddings_images_legacy_trainer_step_vs_oracle():
    """order_embeddings_images.py API surface: Euclidean order energy over precomputed image features."""
    from learning_embeddings_amd import order_embeddings_images as oei
    from test_host_cpu import _fake_loaders
    lm = SyntheticLabelMap([2, 4, 8])
    gd = oe_h.create_combined_graphs(_fake_loaders(lm, 24, 4), lm)
    rs = np.random.RandomState(0)
    fc7 = {'img_%06d' % j: rs.randn(2048).astype(np.float32) for j in range(24)}
    crit = oei.OrderEmbeddingWithImagesLoss(lm, neg_to_pos_ratio=3, alpha=1.0)
    tr = oei.EmbeddingLabelsWithImages(gd, lm, crit, lr=1e-2, batch_size=8, experiment_name='x', embedding_dim=10,
                                       neg_to_pos_ratio=3, image_fc7=fc7, normalize=None, alpha=1.0, has_fixed_alpha=True)
    edges = [(u, v) for u, v in gd['G_train_tc'].edges() if type(v) == str][:8]
    frm = [torch.tensor([u for u, _ in edges])]; to = [[v for _, v in edges]]
    W0 = tr.model.embeddings.weight.detach().cpu().numpy().copy()
    seen = {}
    orig = tr.img_feat_net
    def capturing(names):                                                     # FeatNet output of THIS step (order_embeddings_images.py:143-178)
        out = orig(names); out.retain_grad(); seen['names'] = list(names); seen['out'] = out
        return out
    tr.img_feat_net = capturing
    loss, e_pos, e_neg = tr.train_step(frm, to)
    neg = crit.last_negatives
    N, K, B = lm.n_classes, 3, len(edges)
    assert (neg[:, :K] >= N).all() and (neg[:, K:] < N).all()                 # corrupt image / corrupt label
    # the pinned oracle on the same points (order_embeddings.py:818-824 energy, order_embeddings_images.py:371-470 loss): label
    # rows of the table BEFORE the update, image points = the captured FeatNet outputs, the same negatives
    R = seen['out'].detach().cpu().numpy(); slot = {crit.mapping_from_node_to_ix[n]: j for j, n in enumerate(seen['names'])}
    point = lambda ix: W0[ix] if ix < N else R[slot[ix]]
    fa = [u for u, _ in edges]; ta = [crit.mapping_from_node_to_ix[v] for _, v in edges]
    pf = [fa[b] for b in range(B)] + [fa[b] for b in range(B) for p in range(K)] + [int(neg[b, K + p]) for b in range(B) for p in range(K)]
    pt = [ta[b] for b in range(B)] + [int(neg[b, p]) for b in range(B) for p in range(K)] + [ta[b] for b in range(B) for p in range(K)]
    x = np.stack([point(i) for i in pf]).astype(np.float32); y = np.stack([point(i) for i in pt]).astype(np.float32)
    E = O.order_energy(x, y)
    o_pos = E[:B]; o_neg_v = E[B:B + B * K].reshape(B, K); o_neg_u = E[B + B * K:].reshape(B, K)
    o_neg = np.concatenate([o_neg_v, o_neg_u], axis=1)
    o_loss = float(o_pos.sum() + np.maximum(1.0 - o_neg, 0).sum())
    assert np.abs(e_pos.cpu().numpy() - o_pos).max() <= 1e-4 * max(1.0, np.abs(o_pos).max())
    assert np.abs(e_neg.cpu().numpy().reshape(B, 2 * K) - o_neg).max() <= 1e-4 * max(1.0, np.abs(o_neg).max())
    assert abs(loss.item() - o_loss) <= 1e-4 * max(1.0, abs(o_loss))
    gE = np.concatenate([np.ones(B), -((1.0 - o_neg_v) >= 0).reshape(-1).astype(np.float64), -((1.0 - o_neg_u) >= 0).reshape(-1).astype(np.float64)])
    gx, gy = O.order_energy_grad(x, y, gE)
    gW = np.zeros_like(W0, dtype=np.float64); gR = np.zeros_like(R, dtype=np.float64)
    for ids, gg in ((pf, gx), (pt, gy)):
        for i, row in zip(ids, gg):
            if i < N: gW[i] += row
            else: gR[slot[i]] += row
    assert np.abs(tr.model.embeddings.weight.grad.cpu().numpy() - gW).max() <= 1e-4 * max(1.0, np.abs(gW).max())
    assert np.abs(seen['out'].grad.cpu().numpy() - gR).max() <= 1e-4 * max(1.0, np.abs(gR).max())
    assert (tr.model.embeddings.weight.detach().cpu().numpy() != W0).any()      # the table moved
    l2, _, _ = tr.train_step(frm, to)
    assert torch.isfinite(l2)


def _f12_trainer(tmp_path):
    """The repo's JointEmbeddings on fixture F12's inputs: same loaders, label table, in-memory images and linear stand-in image network
    (followed by FeatCNN18.soft_clip) the reference ran on in tests/golden/make_golden_eval.py."""
    import json
    fx = json.load(open(os.path.join(GOLDEN, 'F12_eval_phase.json')))
    z = np.load(os.path.join(GOLDEN, 'F12_eval_phase.npz'))
    lm = SyntheticLabelMap(fx['levels'])
    names = fx['names']
    images = torch.from_numpy(z['images'])
    dl = {s: [{'level_labels': np.asarray(b['level_labels']), 'image_filename': b['image_filename'],
               'path_to_image': [images[names.index(f)] for f in b['image_filename']]} for b in bl] for s, bl in fx['loaders'].items()}
    gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.01, True, K=fx['K_cone'], use_CNN=True)
    tr = oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=0,
                              batch_size=16, experiment_name='f12', embedding_dim=fx['D'], neg_to_pos_ratio=5, image_fc7=None,
                              normalize=None, alpha=0.01, experiment_dir=str(tmp_path), n_epochs=1, eval_interval=1)
    tr.model.embeddings.weight.data.copy_(torch.from_numpy(z['W']))
    lin_w, lin_b, Kc = torch.from_numpy(z['lin_w']).to(DEV), torch.from_numpy(z['lin_b']).to(DEV), fx['K_cone']

    class Net(torch.nn.Module):
        def forward(self, x):
            return ops.ImageSoftClipFn.apply(x.flatten(1).float() @ lin_w.t() + lin_b, Kc)
    tr.img_feat_net = Net()

    class DS:
        def get_image(self, fname):
            return images[names.index(fname)]
    tr.criterion.set_dataloader(DS())
    return tr, fx, z


def _close(a, b, tol=1e-9):
    if isinstance(b, dict):
        return set(map(str, a)) == set(b) and all(_close(a[k if k in a else int(k)], v, tol) for k, v in b.items())
    return abs(float(a) - float(b)) <= tol


def test_eval_phase_matches_the_reference_run_fixture_f12(tmp_path):
    """calculate_classification_metrics('train' | 'val' | 'test') and check_graph_embedding() (oe_h.py:1971-2247) against the REFERENCE's
    own return values on the same inputs (fixture F12): every entry of the metrics dict incl. `level_metrics`, the two median norms
    (which see the zero rows the reference's chunk loops leave), `image_is_a_member_of`, and the reconstruction 7-tuple.  Counts are
    integers, so the count-derived metrics must agree to the last bit of float64; norms / thresholds to fp32 tolerance.  The
    corrected variant (reference_exact=False: every row embedded) must differ -- the switch does something."""
    tr, fx, z = _f12_trainer(tmp_path)
    assert tr.reference_exact_eval is True                         # the default reproduces the reference
    for phase in ('train', 'val', 'test'):
        want = fx['classification'][phase]
        got = tr.calculate_classification_metrics(phase)
        for key, w in want.items():
            tol = 1e-5 if key.startswith('median') else 1e-12
            assert _close(got[key], w, tol), (phase, key, got[key], w)
    got_tr = tr.calculate_classification_metrics('train')
    assert {str(k): [int(x) for x in v] for k, v in tr.image_is_a_member_of.items()} == fx['image_is_a_member_of']
    assert np.abs(tr.img_rep[0].numpy() - z['img_rep_train']).max() < 1e-5 and not tr.img_rep[0][-1].any()
    best = tr.check_graph_embedding()
    want = fx['reconstruction']
    assert abs(best[1] - want[1]) < 1e-4                           # threshold: an energy value (fp32 kernel against torch CPU)
    for i in (0, 2, 3, 4, 5, 6):
        assert abs(best[i] - want[i]) < 1e-12, (i, best, want)
    fixed = tr.calculate_classification_metrics('val', reference_exact=False)
    assert fixed['median_img_norm'] != got_tr['median_img_norm'] and fixed != fx['classification']['val']
    assert tr.check_graph_embedding(reference_exact=False) != best


def test_classification_metrics_and_reconstruction_vs_bruteforce(tmp_path):
    """calculate_classification_metrics (oe_h.py:1971-2178) and check_graph_embedding (:2180-2247), corrected variant
    (reference_exact=False: every image / label row embedded): the batched GPU versions against a direct per-image / per-pair
    restatement of the reference's loops on the oracle's energies."""
    from test_host_cpu import _fake_loaders
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, 24, 8)
    for split in dl.values():
        for b in split:
            b['path_to_image'] = [torch.rand(3, 32, 32, generator=torch.Generator().manual_seed(int(n[4:]))) for n in b['image_filename']]
    gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.05, True, K=0.1, use_CNN=True)
    tr = oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=0,
                              batch_size=16, experiment_name='m', embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                              normalize=None, alpha=0.05, experiment_dir=str(tmp_path), n_epochs=1, eval_interval=1)
    tr.criterion.set_dataloader(tr.datasets['val'])
    tr.model.eval(); tr.img_feat_net.eval()
    got = tr.calculate_classification_metrics('val', k=[1, 3], reference_exact=False)
    # ---- brute force on the same embeddings
    G = gd['G_val']
    images = [n for n in G if type(n) == str]; labels = sorted(n for n in G if type(n) != str)
    with torch.no_grad():
        img = tr.embed_images(images).cpu().numpy()
        lab = tr.model(torch.arange(lm.n_classes, device=DEV)).cpu().numpy()
    tp = {l: 0 for l in range(lm.n_classes)}; fp = dict(tp); fn = dict(tp); tn = dict(tp); hit = {1: 0, 3: 0}
    for i, name in enumerate(images):
        member = sorted(G.predecessors(name))
        e = O.cone_energy(lab, np.repeat(img[i:i + 1], lm.n_classes, 0), 0.1)
        for lvl in range(3):
            s, t = lm.level_start[lvl], lm.level_stop[lvl]
            order = np.argsort(e[s:t], kind='stable') + s
            for kv in (1, 3):
                hit[kv] += int(member[lvl] in order[:kv])
            if order[0] == member[lvl]:
                tp[member[lvl]] += 1
                for o_ in range(s, t):
                    if o_ != member[lvl]: tn[o_] += 1
            else:
                fp[int(order[0])] += 1; fn[member[lvl]] += 1
    T = {k: sum(d[l] for l in labels) for k, d in (('tp', tp), ('fp', fp), ('fn', fn), ('tn', tn))}
    prec = T['tp'] / max(T['tp'] + T['fp'], 1e-30); rec = T['tp'] / max(T['tp'] + T['fn'], 1e-30)
    f1 = 0.0 if prec + rec == 0 else 2 * prec * rec / (prec + rec)
    assert abs(got['m-f1'] - f1) < 1e-6 and abs(got['accuracy'] - (T['tp'] + T['tn']) / sum(T.values())) < 1e-6
    assert abs(got['hit@1'] - hit[1] / (3 * len(images))) < 1e-6 and abs(got['hit@3'] - hit[3] / (3 * len(images))) < 1e-6
    # ---- reconstruction: best-F1 threshold over all label pairs
    best = tr.check_graph_embedding(reference_exact=False)
    tc = gd['graph_tc']
    pos_pairs = [(u, v) for u, v in tc.edges()]
    nodes = sorted(set(u for u, _ in pos_pairs) | set(v for _, v in pos_pairs))
    E = O.cone_energy(np.repeat(lab[nodes][:, None], len(nodes), 1), np.repeat(lab[nodes][None], len(nodes), 0), 0.1)
    isp = np.zeros((len(nodes), len(nodes)), bool)
    ix = {n: i for i, n in enumerate(nodes)}
    for u, v in pos_pairs: isp[ix[u], ix[v]] = True
    off = ~np.eye(len(nodes), dtype=bool)
    p = E[isp]; n = E[(~isp) & off]
    f1s = []
    for t_ in np.unique(np.concatenate((p, n))):
        cp = (p <= t_).sum(); cn = (n > t_).sum()
        pr = cp / max(cp + (len(n) - cn), 1); rc = cp / len(p)
        f1s.append(0.0 if pr + rc == 0 else 2 * pr * rc / (pr + rc))
    assert abs(best[0] - max(f1s)) < 1e-4


def _dp_engine_worker(rank, world, port, overlap, q, graph=False, chunk=None):
    import os, sys
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), LEC_DIST_BACKEND='gloo')
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import torch as t
    from learning_embeddings_amd.engine import StepEngine as SE
    eng = SE('tiny', n_images=64, dtype='bf16' if chunk is None else 'fp32', overlap_wgrad=overlap, use_graph=graph, graph_after=1, cnn_chunk=chunk)
    assert eng.cnn_chunk == chunk
    negs = []
    for _ in range(3):
        eng.step(); negs.append(eng.last[5].copy())
    t.cuda.synchronize()
    assert ((eng.hip_graph if chunk is None else eng.chunk_graph) is not None) == graph, eng.graph_error
    q.put((rank, eng.arena.data.cpu().numpy(), eng.table.cpu().numpy(), negs))
    t.distributed.barrier(); eng.close(); t.distributed.destroy_process_group()


@pytest.mark.parametrize('overlap,graph,chunk', [(False, False, None), (True, False, None), (True, True, None), (True, False, 8), (True, True, 8)])
def test_step_engine_data_parallel_replicas_stay_identical(overlap, graph, chunk):
    """StepEngine under DP (2 ranks sharing the GPU over gloo), with the shadow-weight / direct-gradient path and with the
    side-stream weight gradients, and with the hipGraph launch mode (all-reduce after the replay): after 3 steps both ranks
    hold bit-identical CNN parameters and label table, and each rank's negatives are its slice of the single-process
    global stream.  chunk=8: the CNN rows in two chunks per step (engine.cnn_chunk, fp32) -- one backward per chunk into the same
    gradient slots: the buckets are reduced ONCE, after the last chunk (a reducer that fires after chunk 0 leaves the later chunks'
    gradients local and the replicas drift apart); with graph=True the chunk is ONE hipGraph replayed per chunk (engine._chunk_body)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    procs = [ctx.Process(target=_dp_engine_worker, args=(r, 2, port, overlap, q, graph, chunk)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda r: r[0])
    for p in procs: p.join(120)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    lm = SyntheticLabelMap([2, 4, 8])
    g = NegativeGraph.from_labelmap(lm, n_images=64, pick_per_level=True, seed=0)
    N, L, B = lm.n_classes, 3, 8
    par = lm.parents()
    for s_ in range(3):
        b = np.arange(2 * B); j = (s_ * 2 * B + b) % 64
        leaf = lm.level_start[-1] + j % lm.levels[-1]
        frm = []
        for bb, lf in zip(b, leaf):
            chain = [lf]
            while chain[-1] in par: chain.append(par[chain[-1]][0])
            frm.append(chain[::-1][bb % L])
        want = g.draw_batch(np.array(frm, dtype=np.int32), (N + j).astype(np.int32), 4)
        assert np.array_equal(np.concatenate([res[0][3][s_], res[1][3][s_]]), want)


# ------------------------------------------------------------------------------------------------ oe.py (Euclidean cones)
def test_euclidean_cones_call_site_vs_reference_fixture():
    """network/oe.py's criterion / Embedder call site (K = 3.0) against fixture F10."""
    from learning_embeddings_amd import oe, _lib
    f = np.load(os.path.join(GOLDEN, 'F10_euclidean_cone.npz')); K = float(f['K'])
    g = lambda k: f['c_' + k]
    lm = SyntheticLabelMap(g('levels').tolist(), edges=[tuple(e) for e in g('edges').tolist()])
    N, M, Kn = lm.n_classes, int(g('n_images')), int(g('Kneg'))
    names = ['img_%06d' % j for j in range(M)]
    n2i = {i: i for i in range(N)}; n2i.update({names[j]: N + j for j in range(M)}); i2n = {v: k for k, v in n2i.items()}
    crit = oe.EuclideanConesWithImagesHypernymLoss(lm, Kn, {}, float(g('alpha')), True, K=K, use_CNN=True)
    crit.set_negative_graph(NegativeGraph.from_labelmap(lm, n_images=M, pick_per_level=True, seed=0), n2i, i2n)
    R = torch.tensor(g('R'), device=DEV, requires_grad=True)

    class DL:
        def get_image(self, fname):
            return R[n2i[fname] - N]

    class Net(torch.nn.Module):                                             # identity "CNN" + oe.py:235-240 soft_clip
        def __init__(self):
            super().__init__(); self.K = K
        def forward_raw(self, x):
            return x.float()
        def forward(self, x):
            return ops.ImageSoftClipFn.apply(x.float(), self.K, _lib.IMAGE_SOFTCLIP_K)
    crit.set_dataloader(DL())
    model = oe.Embedder(g('W').shape[1], lm, None, K=K).to(DEV)
    with torch.no_grad():
        model.embeddings.weight.copy_(torch.tensor(g('W')))
    of = [i2n[int(i)] for i in g('from')]; ot = [i2n[int(i)] for i in g('to')]
    inputs_to = [R[n2i[t] - N] if isinstance(t, str) else t for t in ot]
    crit.seed_sampler(0)
    loss, e_pos, e_neg = crit(model, Net(), list(of), inputs_to, of, ot, torch.ones(len(of)), 'train')
    assert np.array_equal(crit.last_negatives, g('neg'))
    loss.backward()
    assert np.abs(e_pos.detach().cpu().numpy() - g('e_pos')).max() <= 1e-4
    assert np.abs(e_neg.detach().cpu().numpy() - g('e_neg')).max() <= 1e-4
    assert abs(loss.item() - float(g('loss'))) <= 1e-4 * abs(float(g('loss')))
    assert np.abs(model.embeddings.weight.grad.cpu().numpy() - g('gW')).max() / np.abs(g('gW')).max() < 1e-3
    assert np.abs(R.grad.cpu().numpy() - g('gR')).max() / np.abs(g('gR')).max() < 1e-3
    # stand-alone modules: Embedder.forward, E_operator, eval branch
    out = model(torch.tensor(f['emb_idx'][:50], device=DEV))
    assert np.abs(out.detach().cpu().numpy() - O.soft_clip_add(g('W')[f['emb_idx'][:50]], K)).max() < 5e-6
    E = crit.E_operator(torch.tensor(f['x10'], device=DEV), torch.tensor(f['y10'], device=DEV))
    assert (np.abs(E.cpu().numpy() - f['E10']) <= np.maximum(1e-4, 4 * np.abs(f['E10'] - f['E64_10']))).all()


def test_euclidean_cones_trainer_runs(tmp_path):
    from test_host_cpu import _fake_loaders
    from learning_embeddings_amd import oe
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, 32, 8)
    for split in dl.values():
        for b in split:
            b['path_to_image'] = [torch.rand(3, 32, 32, generator=torch.Generator().manual_seed(int(n[4:]))) for n in b['image_filename']]
    gd = oe.create_combined_graphs(dl, lm, pick_per_level=True)
    crit = oe.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 1.0, True, K=3.0, use_CNN=True)
    tr = oe.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-2, n_workers=0,
                            batch_size=16, experiment_name='e', embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                            normalize=None, alpha=1.0, experiment_dir=str(tmp_path), n_epochs=2, eval_interval=1)
    assert isinstance(tr.model, oe.Embedder) and isinstance(tr.img_feat_net, oe.FeatCNN18)
    w0 = tr.model.embeddings.weight.detach().clone()
    tr.run_model()
    assert np.isfinite(tr.last_epoch_loss)
    w1 = tr.model.embeddings.weight.detach()
    assert (w1 - w0).abs().max().item() > 0                                  # plain Adam moved the table ...
    assert w1.norm(dim=1).max().item() > 1.0                                 # ... and nothing clipped it into the unit ball
    assert set(['m-f1', 'hit@1']).issubset(tr.last_metrics)


def _dp_world_size_one_worker(q):
    import ctypes as C                                                       # noqa: F401
    import torch as t
    from learning_embeddings_amd import _lib
    from learning_embeddings_amd.parallel import DpComm
    try:
        dev = t.device('cuda', 0)
        comm = DpComm()
        for dt in (t.float32, t.bfloat16):
            x = t.randn(1 << 20, device=dev).to(dt); ref = x.clone()
            side = t.cuda.Stream(); side.wait_stream(t.cuda.current_stream())
            comm.allreduce_sum_(x, stream=side)
            t.cuda.current_stream().wait_stream(side); t.cuda.synchronize()
            assert t.equal(x, ref)
        try:
            comm.allreduce_sum_(t.zeros(4, dtype=t.int32, device=dev))
            raise AssertionError('an int32 buffer was accepted')
        except ValueError:
            pass
        comm.close()
        rc = _lib.lib.lec_dp_allreduce_sum(None, None, 4, 0, None)
        assert rc == _lib.E_STATE and b'not initialised' in _lib.lib.lec_last_error()
        q.put('ok')
    except BaseException as e:                                               # noqa: BLE001
        q.put('%s: %s' % (type(e).__name__, e))


def test_dp_c_abi_rccl_layer_world_size_one():
    """include/lecone.h (5b): lec_dp_unique_id / lec_dp_init / lec_dp_allreduce_sum / lec_dp_destroy over RCCL, one rank: the sum over
    one rank is the identity, in fp32 and bf16, on a side stream, and a destroyed communicator refuses further calls.  In a child
    process, like every test that brings RCCL up: a communicator created and destroyed inside the pytest process between two graph
    captures made the later graph's replay die inside the HIP runtime (hip::Graph::UpdateStreams, rocgdb) for some test orders."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    p = ctx.Process(target=_dp_world_size_one_worker, args=(q,)); p.start()
    assert q.get(timeout=300) == 'ok'
    p.join(60)


def _dp_lecone_worker(q):
    import os
    os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', LEC_FORCE_DIST='1', LEC_DP_BACKEND='lecone')
    import torch as t
    from learning_embeddings_amd.engine import StepEngine
    eng = StepEngine('tiny', n_images=64, dtype='fp32', use_graph=True, graph_after=2, passes=1)     # one pass: reproducible gradients
    assert eng.reducer.enabled and eng.reducer.comm is not None
    ls = [float(eng.step()) for _ in range(4)]
    t.cuda.synchronize()
    q.put((ls, eng.table.cpu().numpy(), eng.hip_graph is not None))
    eng.close()
    t.distributed.destroy_process_group()


def test_step_engine_gradient_exchange_through_the_c_abi_rccl_layer():
    """LEC_DP_BACKEND=lecone: the engine's bucketed gradient all-reduce (eager buckets and the after-replay reduction) goes through
    lec_dp_allreduce_sum; with one forced rank the step must equal the plain single-process step."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    p = ctx.Process(target=_dp_lecone_worker, args=(q,)); p.start()
    ls, table, graphed = q.get(timeout=300); p.join(60)
    ref = StepEngine('tiny', n_images=64, dtype='fp32', use_graph=True, graph_after=2, passes=1)
    ls_ref = [float(ref.step()) for _ in range(4)]
    torch.cuda.synchronize()
    assert graphed and np.allclose(ls, ls_ref, rtol=1e-5)
    assert np.abs(table - ref.table.cpu().numpy()).max() < 1e-6
    ref.close()


def test_bench_gpus_2_self_launches_on_one_gpu():
    """The driver's plain command for N > 1, `python bench.py --gpus 2 ...` with no torchrun environment: bench.py starts the two ranks
    itself (children of `python -m torch.distributed.run`), they share this box's single GPU (so the gradient exchange runs over gloo:
    the launcher says so), and rank 0 prints ONE JSON line that reports the group it actually reduced over and identical replicas."""
    import json, subprocess, sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '4', '--workload', 'tiny',
                        '--secondary', 'none', '--through-trainer', '0', '--no-stress', '--no-cpu-baseline'],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['steps'] == 3 and out['scaling'] == 'weak'
    dp = out['data_parallel']
    assert dp['world_size'] == 2 and dp['replicas_identical_after_run'] is True and len(dp['buckets']) >= 2
    assert out['config']['global_batch'] == 2 * 8 and out['value'] > 0
    assert out['library_conv_launches_per_step'] == 0 and out['library_conv_launches_whole_run'] == 0, 'no convolution of a bench run may go to a library, in the step or around it'


def test_step_engine_concurrent_half_batch_passes_equal_the_same_passes_in_turn():
    """engine `passes=2` (the fp32 default): the step's CNN rows go through the backbone as two concurrent parts (positives' images |
    image negatives), one HIP stream each, BatchNorm statistics per part.  Must equal the same two parts pushed through IN TURN on
    one stream by plain autograd calls -- raw outputs (deterministic forward kernels; one fully connected GEMM over all rows against two), loss, label-table update, the
    parameter gradients up to float-atomic summation order, and the BatchNorm running statistics (updated in part order: positives,
    then negatives) -- and differ from the one-pass step (whose BatchNorm batch is all rows)."""
    a = StepEngine('tiny', n_images=64, dtype='fp32')
    assert a.passes == 2 and a.overlap.side is None
    b = StepEngine('tiny', n_images=64, dtype='fp32', passes=1, overlap_wgrad=False)
    c = StepEngine('tiny', n_images=64, dtype='fp32', passes=1, overlap_wgrad=False)
    b.img_feat_net.load_state_dict(a.img_feat_net.state_dict()); c.img_feat_net.load_state_dict(a.img_feat_net.state_dict())
    h = b.n_rows // 2

    def core_in_turn(ev=None):                                   # reference: the two parts one after the other, one stream
        codes = b.codes_dev
        from learning_embeddings_amd.engine import _gather_images
        images = _gather_images(b.pool, b.idx_dev)
        b.arena.zero_grad(); b.table_grad.zero_(); b.gfeat.zero_()
        from learning_embeddings_amd import _lib
        b.backbone.bn_grad_accumulate = True                        # two backward passes add into b's gradient slots (engine b's OWN setting: a's is not seen here)
        b.backbone.conv_schedule = _lib.SCHEDULE_TILE_WALK          # the kernels engine a's passes run (a multi-pass step uses the tile walk): same summation order
        parts = [b.img_feat_net.forward_raw(images[p * h:(p + 1) * h]) for p in range(2)]
        feats = torch.cat([f.detach() for f in parts]); b.last_feats = feats
        out = ops.joint_loss_raw(b.table, feats, codes[:, 0].contiguous(), codes[:, 1].contiguous(), codes[:, 2:].contiguous(), None, b.K_cone, b.alpha,
                                 _lib.ENERGY_HYP_CONE, _lib.LABEL_HYP, _lib.IMAGE_SOFTCLIP, b.table_grad, b.gfeat)
        for p in range(2):
            parts[p].backward(b.gfeat[p * h:(p + 1) * h])
        return out
    b._core = core_in_turn
    for s_ in range(2):
        if s_:                                                       # every step starts from ONE state: the float-atomic sums of the previous
            for e in (b, c):                                         # step's gradients differ in their last bits, and Adam's early steps turn that
                e.arena.data.copy_(a.arena.data); e.table.copy_(a.table)     # into +-lr on weights whose gradient is noise
                e.arena.exp_avg.copy_(a.arena.exp_avg); e.arena.exp_avg_sq.copy_(a.arena.exp_avg_sq)
                e.table_m.copy_(a.table_m); e.table_v.copy_(a.table_v)
                e.img_feat_net.load_state_dict(a.img_feat_net.state_dict())
        la = a.step(); lb = b.step(); lc = c.step()
        torch.cuda.synchronize()
        assert torch.allclose(a.last_feats, b.last_feats, rtol=1e-5, atol=1e-6), s_   # (the fully connected GEMM sees 16 rows here, 8 + 8 there)
        assert not torch.equal(a.last_feats, c.last_feats)          # one BatchNorm batch of all rows is a different function
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(la)))
        d = (a.arena.grad - b.arena.grad).double().norm().item() / b.arena.grad.double().norm().item()
        assert d < 1e-5, d
        assert (a.table - b.table).abs().max().item() < 1e-6
        bufs_a = dict(a.img_feat_net.named_buffers()); bufs_b = dict(b.img_feat_net.named_buffers())
        for k in bufs_a:
            if 'running' in k:
                assert torch.allclose(bufs_a[k], bufs_b[k], rtol=1e-6, atol=1e-7), k
    a.close(); b.close(); c.close()


def test_joint_embeddings_trainer_two_concurrent_cnn_passes_equal_two_passes_in_turn(tmp_path):
    """JointEmbeddings.train_step at fp32 pushes the step's CNN batch through the backbone as two concurrent halves
    (_ImageNetBase.forward_raw with cnn_passes = 2: one HIP stream per half, autograd runs each half's backward on its stream).  One
    step must equal the same two halves pushed through one after the other on one stream: loss, label table, the image network's
    gradient up to float-atomic order -- and differ from the one-pass step."""
    from test_host_cpu import _fake_loaders
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, 32, 8)
    for split in dl.values():
        for b in split:
            b['path_to_image'] = [torch.rand(3, 64, 64, generator=torch.Generator().manual_seed(int(n[4:]))).to(DEV) for n in b['image_filename']]
    gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)
    out = {}
    for tag in ('concurrent', 'in_turn', 'one_pass'):
        crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.05, True, K=0.1, use_CNN=True)
        tr = oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-3, n_workers=0,
                                  batch_size=16, experiment_name=tag, embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                                  normalize=None, alpha=0.05, experiment_dir=str(tmp_path), n_epochs=1, eval_interval=5,
                                  cnn_passes=1 if tag == 'one_pass' else 2)
        assert tr.cnn_passes == (1 if tag == 'one_pass' else 2) and (tr.overlap.side is None) == (tag != 'one_pass')
        if tag == 'in_turn':                                     # the same two halves, one stream
            net = tr.img_feat_net
            def in_turn(x, split=None, _n=net):
                h = split if (split is not None and 0 < split < x.shape[0]) else -(-x.shape[0] // 2)   # the criterion's cut: positives | negatives
                return _n.model.fc(torch.cat([_n.model(x[:h], pooled_only=True), _n.model(x[h:], pooled_only=True)])).float()
            net._forward_raw_passes = in_turn
        crit.set_dataloader(tr.datasets['train'])
        tr.train_set.transform = None
        tr.model.train(); tr.img_feat_net.train()
        torch.manual_seed(1)
        it = iter(tr.dataloaders['train'])
        batches = []                                             # rows of every backbone forward of the step = its BatchNorm batches
        hk = tr.img_feat_net.model.conv1.register_forward_hook(lambda m_, i_, o_: batches.append(int(i_[0].shape[0])))
        item = next(it)
        loss = float(tr.train_step(item)[0])
        hk.remove()
        torch.cuda.synchronize()
        if tag == 'concurrent':
            # the two passes are cut at the positives | image-negatives boundary (the reference's own separate forwards,
            # oe_h.py:980-985 | 1003-1009), not at the midpoint of the de-duplicated stack
            n_pos = len({o for o in item['original_to'] if type(o) == str} | {o for o in item['original_from'] if type(o) == str})
            assert batches == [n_pos, crit.last_cnn_rows - n_pos], (batches, n_pos, crit.last_cnn_rows)
        out[tag] = (loss, tr.model.embeddings.weight.detach().clone(), tr.arena.grad.clone(), crit.last_cnn_rows)
    a, b, c = out['concurrent'], out['in_turn'], out['one_pass']
    assert a[3] == b[3] == c[3] and a[3] >= 16
    assert abs(a[0] - b[0]) <= 1e-6 * max(1.0, abs(b[0])) and (a[1] - b[1]).abs().max().item() <= 1e-6
    assert (a[2] - b[2]).double().norm().item() / b[2].double().norm().item() < 1e-5
    assert abs(a[0] - c[0]) > 1e-6 * max(1.0, abs(c[0]))          # one BatchNorm batch of all rows is a different function


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_two_trainers_on_two_threads_with_different_settings_equal_each_alone(tmp_path, dtype):
    """No process-wide switches on the step path: the BatchNorm `accumulate` flag and the convolution
    `schedule` are per-call arguments of the C ABI fed from the owning backbone (ResNet.bn_grad_accumulate / conv_schedule / wgrad_overlap ->
    the FusionContext of each forward), and the context stack is per thread.  Trainer A (two concurrent CNN passes: BatchNorm gradients
    ADD, tile-walk convolutions, weight gradients in line) and trainer B (one pass: BatchNorm gradients OVERWRITE, balanced convolutions
    wherever they apply, weight gradients on a side stream) step at the same time from two Python threads; each must do what it does alone.
    With a shared switch A's BatchNorm gradients would lose a pass (or B's would pile up): far outside the tolerance below.
    dtype = 'bf16' (round 5): the same on the bf16 convolution stack (csrc/conv_mfma.hip, whose one-time kernel-attribute flags used to be plain statics: two
    threads reaching a kernel's first launch together raced on them) -- trainer A one pass with side-stream weight gradients, trainer B the reference's own
    batches (several forwards per step, BatchNorm gradients ADD)."""
    import threading
    from test_host_cpu import _fake_loaders
    from learning_embeddings_amd import _lib
    lm = SyntheticLabelMap([2, 4, 8])
    dl = _fake_loaders(lm, 32, 8)
    for split in dl.values():
        for b in split:
            b['path_to_image'] = [torch.rand(3, 64, 64, generator=torch.Generator().manual_seed(int(n[4:]))).to(DEV) for n in b['image_filename']]
    n_steps = 3

    bf16 = dtype == 'bf16'
    # (the bf16 stack is not run-to-run bit-reproducible even for ONE trainer alone -- the library kernels behind some of its layers sum in a varying order: the same
    # first step gives losses 25.7758 / 25.7756 / 25.7747 in three fresh processes -- so its bars are noise bars; a lost or doubled BatchNorm pass is a factor, not a percent)
    tol_g, tol_l, tol_t = (5e-2, 1e-3, 2.5e-4) if bf16 else (1e-4, 1e-6, 1e-6)

    def build(tag, passes):
        gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)       # its own graphs: the negative sampler (one MT19937 stream) is a trainer's state
        crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, 5, {}, 0.05, True, K=0.1, use_CNN=True)
        kw = dict(compute_dtype=torch.bfloat16, reference_exact_batches=passes == 2) if bf16 else dict(cnn_passes=passes)     # (bf16: "2" = trainer A's role, the settings differ another way)
        tr = oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=1e-4, n_workers=0,
                                  batch_size=16, experiment_name=tag, embedding_dim=10, neg_to_pos_ratio=5, image_fc7=None,
                                  normalize=None, alpha=0.05, experiment_dir=str(tmp_path), n_epochs=1, eval_interval=5, **kw)
        bb = tr.img_feat_net.model
        if passes == 1 and not bf16:
            bb.conv_schedule = _lib.SCHEDULE_BALANCED
        assert bb.bn_grad_accumulate == (passes == 2) and bb.wgrad_overlap is tr.overlap
        crit.set_dataloader(tr.datasets['train']); tr.train_set.transform = None
        tr.model.train(); tr.img_feat_net.train()
        items = []
        it = iter(tr.dataloaders['train'])
        for _ in range(n_steps):
            items.append(next(it))
        return tr, items

    def snapshot(tr):
        return {'net': {k: v.clone() for k, v in tr.img_feat_net.state_dict().items()}, 'table': tr.model.embeddings.weight.detach().clone()}

    def restore(tr, snap):
        """Every step starts from ONE state: the float-atomic sums of a step's weight gradients differ in their last bits from run to run, and
        Adam's first steps turn that into +-lr on every weight whose gradient is noise -- a later step's loss then moves by a percent."""
        tr.img_feat_net.load_state_dict(snap['net']); tr.model.embeddings.weight.data.copy_(snap['table'])
        tr.arena.refresh_lowp()                                              # bf16: the shadow weights the convolutions read follow the restored master (as load_model does)
        for t in (tr.arena.exp_avg, tr.arena.exp_avg_sq, tr.table_m, tr.table_v):
            if t is not None:
                t.zero_()
        tr.table_step = 0; tr.arena.step = 0

    def run(tr, items, out, barrier=None):
        try:
            snap = snapshot(tr)
            out['loss'], out['grad'], out['bn_grad'], out['table'] = [], [], [], []
            for k, item in enumerate(items):
                restore(tr, snap)
                if barrier is not None:
                    barrier.wait(timeout=60)
                loss = tr.train_step(item)[0]
                out['grad'].append(tr.arena.grad.clone())
                bn = [p.grad for n_, p in tr.img_feat_net.named_parameters() if 'bn' in n_ and p.grad is not None]
                out['bn_grad'].append(torch.cat([g.flatten() for g in bn]).clone())
                out['table'].append(tr.model.embeddings.weight.detach().clone())
                out['loss'].append(loss)
            torch.cuda.synchronize()
            out['loss'] = [float(l) for l in out['loss']]
        except Exception as e:                                  # noqa: BLE001  (surface it in the main thread)
            out['error'] = e

    alone = {}
    for tag, passes in (('A', 2), ('B', 1)):
        tr, items = build(tag + '_alone', passes)
        alone[tag] = {}
        run(tr, items, alone[tag])
        assert 'error' not in alone[tag], alone[tag].get('error')
    # bf16: the noise floor is measured, not assumed -- each trainer ALONE a second time; two threads may differ from alone by three times what alone differs from alone
    floor = {'A': 0.0, 'B': 0.0}
    if bf16:
        for tag, passes in (('A', 2), ('B', 1)):
            tr, items = build(tag + '_alone2', passes)
            again = {}
            run(tr, items, again)
            assert 'error' not in again, again.get('error')
            relf = lambda a, b: (a - b).double().norm().item() / max(b.double().norm().item(), 1e-30)
            floor[tag] = max(max(relf(again['bn_grad'][k], alone[tag]['bn_grad'][k]), relf(again['grad'][k], alone[tag]['grad'][k])) for k in range(n_steps))
    both = {'A': {}, 'B': {}}
    trA, itA = build('A_thread', 2); trB, itB = build('B_thread', 1)
    bar = threading.Barrier(2)
    ths = [threading.Thread(target=run, args=(trA, itA, both['A'], bar)), threading.Thread(target=run, args=(trB, itB, both['B'], bar))]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    rel = lambda a, b: (a - b).double().norm().item() / max(b.double().norm().item(), 1e-30)
    for tag in ('A', 'B'):
        got, want = both[tag], alone[tag]
        assert 'error' not in got, got.get('error')
        for k in range(n_steps):
            assert abs(got['loss'][k] - want['loss'][k]) <= tol_l * max(1.0, abs(want['loss'][k])), (tag, k)
            tg = max(tol_g, 3.0 * floor[tag])
            assert rel(got['bn_grad'][k], want['bn_grad'][k]) < tg, (tag, k, rel(got['bn_grad'][k], want['bn_grad'][k]), floor[tag])
            assert rel(got['grad'][k], want['grad'][k]) < tg, (tag, k, floor[tag])
            assert (got['table'][k] - want['table'][k]).abs().max().item() <= tol_t, (tag, k)
    # and the two really ran differently: A's BatchNorm batch is a pass, B's all rows
    assert abs(alone['A']['loss'][0] - alone['B']['loss'][0]) > 1e-6


def test_reference_exact_batches_train_step_matches_the_reference_run_fixture_f13(tmp_path):
    """Fixture F13 (tests/golden/make_golden_step.py, generated by IMPORTING the reference): one whole train step of the reference's joint
    trainer -- criterion.forward's train branch with its FOUR separate CNN forwards (every fixed image end embedded K more times, unflipped;
    each forward its own BatchNorm batch), loss.backward(), the lambda-rescale, ONE Adam over table + CNN at lr, the clip (oe_h.py:904-967,
    980-985, 1003-1009, 1523, 1766-1771) -- on a BatchNorm-bearing CNN (a ResNet-10 of width 8).  Here: JointEmbeddings.train_step with
    reference_exact_batches=True on the same network built from THIS package's modules (liblecone's convolution / BatchNorm / pooling /
    loss / optimizer kernels).  Negatives bit-exact; loss and energies 1e-4; table 2e-6; BatchNorm running statistics 1e-5; gradients 1e-3 of
    their scale; parameters after Adam: a first Adam step moves every weight by lr * g / (|g| + 1e-8), so an entry is compared to 2 % of lr
    where its gradient is above 1e-6 (below that the quotient amplifies float noise of the gradient itself)."""
    from test_host_cpu import _fake_loaders
    from learning_embeddings_amd.resnet import ResNet, BasicBlock
    z = np.load(os.path.join(GOLDEN, 'F13_train_step.npz'))
    lm = SyntheticLabelMap([int(v) for v in z['levels']])
    assert sorted(lm.edges) == [tuple(int(a) for a in e) for e in z['edges']]
    n_img, Kc, D, Kneg, lr = int(z['n_images']), float(z['K']), int(z['D']), int(z['Kneg']), float(z['lr'])
    imgs = torch.from_numpy(z['images_u8']).permute(0, 3, 1, 2).contiguous().float().div(255)
    dl = _fake_loaders(lm, n_img, 8)
    for split in dl.values():
        for b in split:
            b['path_to_image'] = [imgs[int(n[4:]) % n_img] for n in b['image_filename']]
    gd = oe_h.create_combined_graphs(dl, lm, pick_per_level=True)
    N = lm.n_classes
    assert gd['mapping_ix_to_node'][N + 3] == 'img_000003'
    crit = oe_h.EuclideanConesWithImagesHypernymLoss(lm, Kneg, {}, float(z['alpha']), True, K=Kc, use_CNN=True)
    net = oe_h.FeatCNN18(image_dir='', output_dim=D, K=Kc, arch=lambda: ResNet(BasicBlock, [1, 1, 1, 1], width=int(z['width'])))
    tr = oe_h.JointEmbeddings(gd, dl, image_dir='', use_CNN=True, labelmap=lm, criterion=crit, lr=lr, n_workers=0, batch_size=len(z['from']),
                              experiment_name='f13', embedding_dim=D, neg_to_pos_ratio=Kneg, image_fc7=None, normalize=None, alpha=float(z['alpha']),
                              experiment_dir=str(tmp_path), n_epochs=1, eval_interval=5, img_feat_net=net, reference_exact_batches=True)
    assert tr.img_feat_net is net and tr.cnn_passes == 1 and net.model.bn_grad_accumulate
    sd0 = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('sd0/')}
    sd1 = {k[4:]: z[k] for k in z.files if k.startswith('sd1/')}
    assert set(sd0) == set(net.state_dict())                        # torchvision's key names on both sides
    net.load_state_dict(sd0)
    tr.model.embeddings.weight.data.copy_(torch.from_numpy(z['W0']))
    i2n = gd['mapping_ix_to_node']
    of = [i2n[int(i)] for i in z['from']]; ot = [i2n[int(i)] for i in z['to']]
    item = {'from': list(of), 'to': [(imgs[int(i) - N].flip(-1) if f else imgs[int(i) - N]) if int(i) >= N else int(i) for i, f in zip(z['to'], z['flips'])],
            'status': torch.ones(len(of)), 'original_from': of, 'original_to': ot}
    crit.set_dataloader(tr.datasets['train']); tr.train_set.transform = None
    tr.model.train(); tr.img_feat_net.train()
    crit.seed_sampler(0)
    loss, e_pos, e_neg = tr.train_step(item)
    torch.cuda.synchronize()
    assert np.array_equal(crit.last_negatives, z['neg'])
    assert crit.last_cnn_rows == 8 + int((z['neg'] >= N).sum()) + Kneg * 8      # positives' images + image negatives + K re-embeds of every fixed image end
    assert abs(float(loss) - float(z['loss'])) <= 1e-4 * max(1.0, abs(float(z['loss'])))
    assert np.abs(e_pos.detach().cpu().numpy().reshape(-1) - z['e_pos'].reshape(-1)).max() <= 1e-4
    assert np.abs(e_neg.detach().cpu().numpy().reshape(-1) - z['e_neg'].reshape(-1)).max() <= 1e-4
    assert np.abs(tr.model.embeddings.weight.detach().cpu().numpy() - z['W1']).max() <= 2e-6
    got = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    params = dict(net.named_parameters())
    for k, want in sd1.items():
        if 'running_' in k:
            assert np.abs(got[k] - want).max() <= 1e-5, k               # four forwards, four momentum updates, in the reference's order
        elif 'num_batches_tracked' in k:
            continue                                                    # (not advanced here: it only matters for momentum=None, resnet.py)
        else:
            g_ref = z['grad/' + k]; g = params[k].grad.detach().cpu().numpy()
            scale = np.abs(g_ref).max()
            assert np.abs(g - g_ref).max() <= 1e-3 * scale + 1e-7, (k, np.abs(g - g_ref).max(), scale)
            big = np.abs(g_ref) > 1e-6
            assert np.abs(got[k] - want)[big].max(initial=0.0) <= 2e-2 * lr, (k, np.abs(got[k] - want)[big].max(initial=0.0))
            assert np.abs(got[k] - want).max() <= 2.0 * lr + 1e-7, k    # nobody moves further than an Adam step
