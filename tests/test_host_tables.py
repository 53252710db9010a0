"""Host-side controller tables (integer / index work): the product's functions must reproduce the reference's tables
bit-for-bit.  tests/golden/host_tables.json was produced by the reference's own models/p2p/seq_aligner.py,
utils/utils.py and attention_control.py (oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from pnpinversion_amd.p2p import attention_control as ac
from pnpinversion_amd.p2p import seq_aligner
from pnpinversion_amd.text import WordTokenizer
from pnpinversion_amd.utils.utils import get_time_words_attention_alpha, get_word_inds

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "host_tables.json")))


@pytest.mark.parametrize("e", G, ids=[e["src"][:20] for e in G])
def test_tables_match_reference(e):
    tok = WordTokenizer()
    src, tgt, (w0, w1) = e["src"], e["tgt"], e["blend"]
    assert get_word_inds(src, w0, tok).tolist() == e["word_inds_src"]
    assert get_word_inds(tgt, w1, tok).tolist() == e["word_inds_tgt"]
    assert get_word_inds(tgt, 1, tok).tolist() == e["word_inds_int"]
    m, a = seq_aligner.get_refinement_mapper([src, tgt], tok)
    assert m[0].tolist() == e["refine_mapper"]
    assert a[0].tolist() == e["refine_alphas"]
    if "replace_mapper" in e:
        assert seq_aligner.get_replacement_mapper([src, tgt], tok)[0].tolist() == e["replace_mapper"]
    else:
        with pytest.raises(ValueError, match="attention replacement edit can only be applied on prompts with the same length"):
            seq_aligner.get_replacement_mapper([src, tgt], tok)
    for steps in (50, 2):
        al = get_time_words_attention_alpha([src, tgt], steps, {"default_": 0.4}, tok)
        assert al.reshape(steps + 1, 77).tolist() == e["cross_alpha_%d" % steps]
    assert ac.get_equalizer(tgt, (w1,), (2,), tokenizer=tok)[0].tolist() == e["equalizer"]
    lb = ac.LocalBlend([src, tgt], ((w0,), (w1,)), tokenizer=tok, num_ddim_steps=50)
    assert lb.alpha_layers.reshape(2, 77).tolist() == e["lb_alpha"]
    assert lb.start_blend == e["lb_start"]


def test_controller_descriptor_semantics():
    """The declarative tables reproduce the reference's per-row edit formulas (attention_control.py:269-345) exactly."""
    import torch
    from types import SimpleNamespace
    tok = WordTokenizer()
    e = G[0]
    prompts = [e["src"], e["tgt"]]
    pipe = SimpleNamespace(tokenizer=tok)
    rng = np.random.default_rng(0)
    base = torch.from_numpy(rng.random((8, 5, 77), dtype=np.float32))
    repl = torch.from_numpy(rng.random((1, 8, 5, 77), dtype=np.float32))
    for is_replace, blend, eq in ((False, ((e["blend"][0],), (e["blend"][1],)), {"words": (e["blend"][1],), "values": (2,)}),
                                  (True, None, None), (False, None, None)):
        c = ac.make_controller(pipe, prompts, is_replace, {"default_": 0.4}, 0.6, blend, eq, num_ddim_steps=50)
        t = c.tables()
        assert t.self_range == (0, 30) and t.cross_alpha.shape == (51, 77)
        mm, al, eqv = torch.from_numpy(t.mapper), torch.from_numpy(t.alphas), torch.from_numpy(t.equalizer)
        for step in (0, 19, 20, 50):
            a_t = torch.from_numpy(t.cross_alpha[step])
            # reference formulation
            if is_replace:
                new = torch.einsum("hpw,bwn->bhpn", base, torch.from_numpy(np.asarray(e["replace_mapper"], dtype=np.float32))[None])
            else:
                idx = torch.tensor(e["refine_mapper"])
                ra = torch.tensor(e["refine_alphas"]).reshape(1, 1, 1, 77)
                new = base[:, :, idx[None]].permute(2, 0, 1, 3) * ra + repl * (1 - ra)
            if eq is not None:
                new = new * torch.tensor(e["equalizer"]).reshape(1, 1, 1, 77)
            ref = new * a_t + (1 - a_t) * repl
            # declarative formulation consumed by the HIP kernel
            c1 = a_t * eqv * al
            c2 = a_t * eqv * (1 - al) + (1 - a_t)
            got = c1 * torch.einsum("hpw,wn->hpn", base, mm)[None] + c2 * repl
            assert torch.allclose(got, ref, atol=1e-6), (is_replace, step)
        if blend is not None:
            assert t.lb_alpha.tolist() == e["lb_alpha"] and t.lb_start == 10 and abs(t.lb_threshold - 0.3) < 1e-7
        else:
            assert t.lb_alpha is None


def test_method_dispatch_matches_reference():
    """P2PEditor.__call__: every one of the reference's 39 method strings goes to the handler, with the method-specific arguments,
    that the reference's own __call__ (models/p2p_editor.py:28-135) uses (tests/golden/method_dispatch.json); anything else raises the
    reference's NotImplementedError message.  All 39 have a native path."""
    import json
    import types
    from pnpinversion_amd.p2p_editor import P2PEditor
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "method_dispatch.json")))
    fake = types.SimpleNamespace(scheduler=types.SimpleNamespace(set_timesteps=lambda n: None))
    ed = P2PEditor(["x"], "cpu", num_ddim_steps=50, pipeline=fake)
    for n in [n for n in dir(P2PEditor) if n.startswith("edit_image")]:
        setattr(ed, n, (lambda n: (lambda *a, **k: (n, k)))(n))
    unbuilt = set()     # every string reaches its handler
    assert len(gold) == 40
    for m, want in gold.items():
        if m == "__unknown__":
            with pytest.raises(NotImplementedError) as ei:
                ed("no-such-method", "x", "a", "b")
            assert str(ei.value) == want
            continue
        call = lambda: ed(m, image_path="x", prompt_src="a", prompt_tar="b", guidance_scale=7.5, cross_replace_steps=0.4,
                          self_replace_steps=0.6, blend_word=None, eq_params=None, proximal="l0", quantile=0.75,
                          use_inversion_guidance=True, recon_lr=1, recon_t=400)
        if m in unbuilt:
            with pytest.raises(NotImplementedError, match="not built"):
                call()
            continue
        handler, kw = call()
        assert handler == want["handler"], (m, handler, want["handler"])
        for k, v in want["kwargs"].items():
            if m == "negative-prompt-inversion+p2p" and k not in ("guidance_scale", "proximal"):
                continue          # the reference forwards the sweep's recon_* / quantile here too; with proximal=None they are inert
            assert k in kw and kw[k] == v, (m, k, kw.get(k), v)


def test_clip_bpe_tokenizer_matches_transformers():
    """text.ClipBPETokenizer vs transformers' CLIPTokenizer on the same (synthetic) vocabulary and merge list: ids, padding,
    truncation, decode of single tokens (what utils.get_word_inds relies on)."""
    transformers = pytest.importorskip("transformers")
    from pnpinversion_amd.text import ClipBPETokenizer, _bytes_to_unicode
    syms = list(_bytes_to_unicode().values())
    vocab = syms + [s + "</w>" for s in syms]
    merges = [("c", "a"), ("ca", "t</w>"), ("d", "o"), ("do", "g</w>"), ("t", "h"), ("th", "e</w>"), ("s", "i"), ("si", "t"), ("sit", "t"),
              ("i", "n"), ("in", "g</w>"), ("sitt", "ing</w>"), ("o", "n</w>"), ("w", "o"), ("wo", "o"), ("woo", "d"), ("wood", "e"),
              ("woode", "n</w>"), ("c", "h"), ("ch", "a"), ("cha", "i"), ("chai", "r</w>"), ("'", "s</w>"), ("!", "!</w>"), ("Ã", "©</w>"), ("c", "a"), ("ca", "f"), ("caf", "Ã©</w>")]
    seen, uniq = set(), []
    for m in merges:
        if m not in seen:
            seen.add(m); uniq.append(m)
    for a, b in uniq:
        vocab.append(a + b)
    vocab += ["<|startoftext|>", "<|endoftext|>"]
    vmap = {t: i for i, t in enumerate(dict.fromkeys(vocab))}
    ref = transformers.CLIPTokenizer(vocab=vmap, merges=[(a, b) for a, b in uniq])
    mine = ClipBPETokenizer(vmap, uniq)
    texts = ["a cat sitting on the wooden chair", "The   CAT's 12 dogs!!", "", "dog", "café on the chair, a (small) cat-dog",
             " ".join(["sitting"] * 100), "naïve ÅNGSTRÖM x²", "it's we're I'd they'll"]
    a = ref(texts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    b = mine(texts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    assert a.shape == b.shape == (len(texts), 77)
    assert (a == b).all(), [(i, a[i][:20].tolist(), b[i][:20].tolist()) for i in range(len(texts)) if not (a[i] == b[i]).all()]
    for t in texts[:5]:
        ia, ib = ref.encode(t), mine.encode(t)
        assert ia == ib
        for tok in ia[1:-1]:
            assert ref.decode([tok]) == mine.decode([tok]), (tok, ref.decode([tok]), mine.decode([tok]))
    # the reference's word-index helper works on it (utils/utils.py:84-102)
    assert get_word_inds("a cat sitting on the wooden chair", "cat", mine).tolist() == get_word_inds("a cat sitting on the wooden chair", "cat", ref).tolist() == [2]


def test_random_prompt_pairs_against_the_live_reference():
    """Where /root/reference is present (the build container; the GPU box has only the committed goldens above): 60 seeded random
    prompt pairs -- substitutions, insertions, deletions, repeated words, different lengths -- through the reference's own
    seq_aligner / get_time_words_attention_alpha / get_equalizer / LocalBlend and through the product's, bit for bit."""
    from oracle import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not on this machine")
    ref_shim.install()
    from models.p2p import seq_aligner as ref_sa
    from models.p2p.attention_control import LocalBlend as RefLB, get_equalizer as ref_eq
    from utils.utils import get_time_words_attention_alpha as ref_alpha, get_word_inds as ref_inds
    import torch
    tok = WordTokenizer()
    vocab = "a the cat dog sitting on wooden red chair photo of mountain snowy watercolor big small tree and with elephant walking".split()
    rng = np.random.default_rng(1234)
    n_replace = 0
    for case in range(60):
        src = [vocab[i] for i in rng.integers(0, len(vocab), rng.integers(3, 12))]
        tgt = list(src)
        for _ in range(rng.integers(1, 4)):                      # 1-3 edits
            op, pos = rng.integers(0, 3), int(rng.integers(0, len(tgt)))
            w = vocab[int(rng.integers(0, len(vocab)))]
            if op == 0:
                tgt[pos] = w
            elif op == 1:
                tgt.insert(pos, w)
            elif len(tgt) > 2:
                del tgt[pos]
        ps, pt = " ".join(src), " ".join(tgt)
        m, a = seq_aligner.get_refinement_mapper([ps, pt], tok)
        rm, ra = ref_sa.get_refinement_mapper([ps, pt], tok)
        assert torch.equal(torch.as_tensor(m), rm) and torch.equal(torch.as_tensor(a), ra), (ps, pt)
        if len(src) == len(tgt):
            n_replace += 1
            assert torch.equal(torch.as_tensor(seq_aligner.get_replacement_mapper([ps, pt], tok)), ref_sa.get_replacement_mapper([ps, pt], tok)), (ps, pt)
        w0, w1 = src[int(rng.integers(0, len(src)))], tgt[int(rng.integers(0, len(tgt)))]
        assert get_word_inds(ps, w0, tok).tolist() == ref_inds(ps, w0, tok).tolist()
        for steps, frac in ((50, 0.4), (7, {"default_": 0.8, w1: (0.1, 0.5)})):
            cr = frac if isinstance(frac, dict) else {"default_": frac}
            got = get_time_words_attention_alpha([ps, pt], steps, cr, tok)
            assert torch.equal(torch.as_tensor(got), ref_alpha([ps, pt], steps, dict(cr), tok)), (ps, pt, steps)
        assert torch.equal(torch.as_tensor(ac.get_equalizer(pt, (w1,), (2.5,), tokenizer=tok)), ref_eq(pt, (w1,), (2.5,), tokenizer=tok))
        with ref_shim.cuda_to_cpu():
            rlb = RefLB([ps, pt], ((w0,), (w1,)), tokenizer=tok, num_ddim_steps=50)
        lb = ac.LocalBlend([ps, pt], ((w0,), (w1,)), tokenizer=tok, num_ddim_steps=50)
        assert torch.equal(torch.as_tensor(lb.alpha_layers).reshape(-1), rlb.alpha_layers.reshape(-1).cpu()) and lb.start_blend == rlb.start_blend
    assert n_replace >= 5


class _FakeEngine:
    """What NativeUNet needs from an engine to build its attention-site markers (no GPU): the model configuration."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.callback = "unset"

    def set_attention_callback(self, fn):
        self.callback = fn


def test_the_references_own_register_attention_control_hooks_the_native_unet():
    """SURVEY 8b level 1 with the reference's UNMODIFIED code (build container only): models/p2p/attention_control.register_attention_control
    walks NativeUNet.named_children(), finds the 32 markers of class `CrossAttention` (:62-81), assigns their .forward -- and the native UNet
    ends up with the controller the closure carries: a kernel descriptor for the reference's own AttentionReplace / Refine / Reweight objects
    (read off their attributes, bit-identical to this package's classes; with LocalBlend -- substruct words included -- the instance's
    step_callback is pointed at the native blend), nothing for controller=None."""
    from oracle import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not on this machine")
    ref_shim.install()
    import types
    import torch
    from models.p2p import attention_control as ref_ac
    from pnpinversion_amd.config import SD1
    from pnpinversion_amd.pipeline import NativeUNet
    tok = WordTokenizer()
    eng = _FakeEngine(SD1)
    model = types.SimpleNamespace(unet=NativeUNet(eng), tokenizer=tok, device="cpu")
    names = [n for n, _ in model.unet.named_children()]
    assert any("down" in n for n in names) and any("mid" in n for n in names) and any("up" in n for n in names)

    # controller=None: the reference's DummyController is counted up to 32 and means "no controller"
    ref_ac.register_attention_control(model, None)
    assert model.unet.controller is None

    ps, pt = "a cat sitting on a wooden chair", "a dog sitting on a wooden chair"
    prompts = [ps, pt]
    with ref_shim.cuda_to_cpu():
        ref_replace = ref_ac.AttentionReplace(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, tokenizer=tok)
        ref_refine = ref_ac.AttentionRefine(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, tokenizer=tok)
        eq = ref_ac.get_equalizer(pt, ("dog",), (2,), tokenizer=tok)
        ref_rw = ref_ac.AttentionReweight(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, equalizer=eq,
                                          controller=ref_refine)          # (the reference's Reweight takes no tokenizer, :347-355)
        ref_lb = ref_ac.AttentionRefine(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, tokenizer=tok,
                                        local_blend=ref_ac.LocalBlend(prompts, (("cat",), ("dog",)), tokenizer=tok, num_ddim_steps=50))
        ref_lb_sub = ref_ac.AttentionRefine(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, tokenizer=tok,
                                            local_blend=ref_ac.LocalBlend(prompts, (("cat",), ("dog",)), substruct_words=(("chair",), ("wooden",)),
                                                                          th=(0.3, 0.45), tokenizer=tok, num_ddim_steps=50))
        ref_store = ref_ac.AttentionStore()
    mine_replace = ac.AttentionReplace(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, tokenizer=tok)
    mine_refine = ac.AttentionRefine(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, tokenizer=tok)
    mine_rw = ac.AttentionReweight(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6,
                                   equalizer=ac.get_equalizer(pt, ("dog",), (2,), tokenizer=tok), controller=mine_refine, tokenizer=tok)

    def same_tables(a, b):
        for k in ("cross_alpha", "mapper", "alphas", "equalizer"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), k
        assert a.self_range == b.self_range and a.lb_alpha is None and b.lb_alpha is None

    for ref_c, mine in ((ref_replace, mine_replace), (ref_refine, mine_refine), (ref_rw, mine_rw)):
        ref_ac.register_attention_control(model, ref_c)
        assert ref_c.num_att_layers == 32                                   # the reference counted 32 sites
        got = model.unet.controller
        assert isinstance(got, ac.ForeignControllerAdapter) and got.wrapped is ref_c and not ac.is_callback_controller(got)
        same_tables(got.tables(), mine.tables())
        got.cur_step += 1                                                   # step bookkeeping lands on the reference's object
        assert ref_c.cur_step == 1
        ref_c.cur_step = 0

    # LocalBlend: the descriptor carries its selectors, and the reference object's step_callback (which would read the never-filled
    # attention_store) is pointed at the native blend on the engine's accumulators; counter / start_blend stay on the reference's object
    class _BlendEngine(_FakeEngine):
        calls = []

        def local_blend(self, x_t, step_index):
            self.calls.append(step_index)
            return x_t
    beng = _BlendEngine(SD1)
    bmodel = types.SimpleNamespace(unet=NativeUNet(beng), tokenizer=tok, device="cpu")
    for ref_c, sub, th in ((ref_lb, None, (0.3, 0.3)), (ref_lb_sub, (("chair",), ("wooden",)), (0.3, 0.45))):
        mine = ac.AttentionRefine(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, tokenizer=tok,
                                  local_blend=ac.LocalBlend(prompts, (("cat",), ("dog",)), substruct_words=sub, th=th, tokenizer=tok,
                                                            num_ddim_steps=50))
        ref_ac.register_attention_control(bmodel, ref_c)
        got = bmodel.unet.controller
        assert isinstance(got, ac.ForeignControllerAdapter) and got.wrapped is ref_c and not ac.is_callback_controller(got)
        a, b = got.tables(), mine.tables()
        for k in ("cross_alpha", "mapper", "alphas", "equalizer", "lb_alpha"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), k
        assert (a.lb_start, a.lb_threshold, a.lb_threshold_sub) == (b.lb_start, b.lb_threshold, b.lb_threshold_sub) == (10, th[0], th[1])
        assert (a.lb_sub_alpha is None) == (sub is None) and (sub is None or np.array_equal(a.lb_sub_alpha, b.lb_sub_alpha))
        if sub is not None:
            assert a.lb_sub_alpha.sum(1).tolist() == [1.0, 1.0] and not np.array_equal(a.lb_sub_alpha[0], a.lb_sub_alpha[1])
        # the reference's loop calls ITS object's step_callback once per step: blend from the 11th call on (counter > start_blend = 10)
        beng.calls.clear()
        x = torch.zeros(2, 4, 8, 8)
        for mine_too in (ref_c, mine):
            if mine_too is mine:
                bmodel.unet.set_controller(mine)
            for _ in range(12):
                x = mine_too.step_callback(x)
        assert beng.calls == [10, 11, 10, 11] and ref_c.local_blend.counter == 12 and mine.local_blend.counter == 12
    with pytest.raises(RuntimeError, match="not registered with a native UNet"):
        ac.AttentionRefine(prompts, 50, cross_replace_steps={"default_": 0.4}, self_replace_steps=0.6, tokenizer=tok,
                           local_blend=ac.LocalBlend(prompts, (("cat",), ("dog",)), tokenizer=tok, num_ddim_steps=50)).step_callback(x)
    # a plain AttentionStore: no edit (its step bookkeeping still runs on the reference's object)
    ref_ac.register_attention_control(model, ref_store)
    assert model.unet.controller.tables() is None and model.unet.controller.wrapped is ref_store
    # this package's own classes through the reference's function: unchanged
    ref_ac.register_attention_control(model, mine_rw)
    assert model.unet.controller is mine_rw and mine_rw.num_att_layers == 32
    # a hook that is not the reference's protocol is refused, not ignored
    site = next(iter(next(iter(model.unet.named_children()))[1].children()))
    with pytest.raises(TypeError, match="close over a `controller`"):
        site.forward = lambda x, context=None, mask=None: x


def test_native_unet_attention_site_markers_without_the_reference():
    """The same protocol with a locally written registration function of the reference's shape (runs everywhere)."""
    import types
    from pnpinversion_amd.config import SD1, SMALL64_LB
    from pnpinversion_amd.pipeline import NativeUNet

    def register(model, controller):           # models/p2p/attention_control.py:12-81, condensed
        def ca_forward(self, place_in_unet):
            def forward(x, context=None, mask=None, **kwargs):
                return controller(x, context is not None, place_in_unet)
            return forward

        def rec(net_, count, place):
            if net_.__class__.__name__ == "CrossAttention":
                net_.forward = ca_forward(net_, place)
                return count + 1
            for ch in net_.children():
                count = rec(ch, count, place)
            return count
        n = 0
        for name, net in model.unet.named_children():
            for place in ("down", "up", "mid"):
                if place in name:
                    n += rec(net, 0, place)
        controller.num_att_layers = n

    class Probe:
        def __call__(self, attn, is_cross, place):
            return attn

    for cfg, want in ((SD1, 32), (SMALL64_LB, 22)):
        eng = _FakeEngine(cfg)
        unet = NativeUNet(eng)
        model = types.SimpleNamespace(unet=unet)
        p = Probe()
        register(model, p)
        assert p.num_att_layers == want == unet.num_att_layers and unet.controller is p and ac.is_callback_controller(p)
        sites = [s for _, cont in unet.named_children() for s in cont.children()]
        assert [s.is_cross for s in sites] == [False, True] * (want // 2) and [s.index for s in sites] == list(range(want))
        first = cfg.block_out_channels[[i for i, h in enumerate(cfg.block_has_attn) if h][0]]
        assert sites[0].heads == cfg.heads and abs(sites[0].scale - (first // cfg.heads) ** -0.5) < 1e-12
        # replacing it with a descriptor controller drops a callback the engine still holds
        unet._cb_for = p
        unet.set_controller(None)
        assert eng.callback is None and unet._cb_for is None


def test_masactrl_descriptor_windows_and_lists():
    """MutualSelfAttentionControl(start_step, start_layer, layer_idx, step_idx) (models/masactrl/masactrl.py:14-37, membership tests at :61)
    -> pnpi_ctrl_desc kind 2: the windows travel as two integers, explicit lists as a block mask (bit 31 = "a list was given") and a
    per-step flag array; host side only, no GPU."""
    from pnpinversion_amd.engine import MasaCtrlTables
    from pnpinversion_amd.masactrl.masactrl import MutualSelfAttentionControl

    def membership(d, block, step):       # what api_graph.inc's masa_layer / masa_step evaluate from the descriptor
        layer = ((d.masa_layer_mask >> block) & 1) == 1 and block < 31 if d.masa_layer_mask else block >= d.masa_start_layer
        if d.masa_n_steps > 0 and bool(d.masa_step_on_host):
            st = 0 <= step < d.masa_n_steps and d.masa_step_on_host[step] == 1
        else:
            st = step >= d.masa_start_step
        return layer and st

    # the default windows: no mask, no step array
    c = MutualSelfAttentionControl(4, 10)
    t = c.tables(); d = t.desc()
    assert (d.kind, d.masa_start_step, d.masa_start_layer, d.masa_layer_mask, d.masa_n_steps) == (2, 4, 10, 0, 0)
    for block in range(16):
        for step in range(50):
            assert membership(d, block, step) == (step in c.step_idx and block in c.layer_idx)
    # lists that ARE the windows stay windows; anything else becomes mask / flags, including the empty lists and entries past the UNet
    assert MutualSelfAttentionControl(4, 10, layer_idx=list(range(10, 16)), step_idx=list(range(4, 50))).tables().desc().masa_layer_mask == 0
    for layer_idx, step_idx in (([0, 3, 15], [0, 7, 49]), ([], [5]), ([12], []), ([2, 40, 31], [60, 1]), (list(range(16)), list(range(50)))):
        c = MutualSelfAttentionControl(4, 10, layer_idx=layer_idx, step_idx=step_idx)
        t = c.tables(); d = t.desc()
        assert d.masa_layer_mask >> 31 == 1 and d.masa_n_steps >= 1
        for block in range(16):
            for step in range(64):
                assert membership(d, block, step) == (step in step_idx and block in layer_idx), (layer_idx, step_idx, block, step)
    # the step array outlives desc(): the descriptor points into the tables object
    t = MasaCtrlTables(step_idx=[3])
    d = t.desc()
    assert d.masa_n_steps == 4 and [d.masa_step_on_host[i] for i in range(4)] == [0, 0, 0, 1]
