"""which piece of the backward pass does ROCm's stream capture / hipGraph instantiate not survive?  one case per process"""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
case = sys.argv[1]
dev = 'cuda'
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()


def cap(fn, warm=True):
    if warm:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn(); fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        out = fn()
    g.replay(); g.replay()
    torch.cuda.synchronize()
    return out


if case == 'zeros':
    cap(lambda: torch.zeros(16 << 20, device=dev))
elif case == 'zeros_big':
    cap(lambda: torch.zeros(64 << 20, device=dev))
elif case == 'aten_bwd':
    w = torch.randn(256, 256, device=dev, requires_grad=True)
    x = torch.randn(64, 256, device=dev)
    cap(lambda: torch.autograd.grad((x @ w).relu().sum(), [w]))
elif case == 'aten_bwd_backward':
    w = torch.randn(256, 256, device=dev, requires_grad=True)
    x = torch.randn(64, 256, device=dev)

    def f():
        w.grad = None
        (x @ w).relu().sum().backward()
    cap(f)
else:
    import brcnn  # noqa: F401
    from brcnn import Config, build_detector, blocks
    from brcnn import autograd as A
    from tests import util
    A.WGRAD_SIDE_STREAM = '_side' in case
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    cfg = Config.fromfile(os.path.join(root, 'configs', 'boosting_rcnn', 'boosting_rcnn_r50_pafpn_1x_coco.py'))
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=10))
    m = m.cuda().train()
    blocks.conv_weights_channels_last(m)
    m.set_compute_dtype('bf16')
    if case.startswith('conv1'):        # one trainable conv (custom Function: dgrad + wgrad + arena)
        conv = m.neck.fpn_convs[0]
        x = torch.randn(2, 16, 24, 256, device=dev).to(torch.bfloat16).requires_grad_()

        def f():
            A.grad_arena.new_step()
            y = conv.forward_nhwc(x)
            r = torch.autograd.grad(y, [x, conv.conv.weight], torch.ones_like(y))
            A.join_side_streams()
            return r
        cap(f)
    elif case.startswith('neck'):
        feats_in = [torch.randn(2, h, w, c, device=dev).to(torch.bfloat16).requires_grad_() for (h, w, c) in
                    ((32, 48, 256), (16, 24, 512), (8, 12, 1024), (4, 6, 2048))]
        params = [p for p in m.neck.parameters() if p.requires_grad]

        def f():
            A.grad_arena.new_step()
            outs = m.neck.forward_nhwc(feats_in)
            r = torch.autograd.grad(outs, params, [torch.ones_like(o) for o in outs], allow_unused=True)
            A.join_side_streams()
            return r
        cap(f)
    elif case.startswith('layer'):
        layer = m.backbone.layer3
        x = torch.randn(2, 16, 24, 512, device=dev).to(torch.bfloat16).requires_grad_()
        params = [p for p in layer.parameters() if p.requires_grad]

        def f():
            A.grad_arena.new_step()
            y = layer.forward_nhwc(x)
            r = torch.autograd.grad(y, params, torch.ones_like(y), allow_unused=True)
            A.join_side_streams()
            return r
        cap(f)
    elif case.startswith('trunk1'):      # whole trunk, forward + backward in ONE capture
        img = torch.randn(2, 3, 128, 192, device=dev)
        params = [p for mm in (m.backbone, m.neck) for p in mm.parameters() if p.requires_grad]

        def f():
            A.grad_arena.new_step()
            outs = m.extract_feat_nhwc(img)
            r = torch.autograd.grad(outs, params, [torch.ones_like(o) for o in outs], allow_unused=True)
            A.join_side_streams()
            return r
        cap(f)
    elif case.startswith('two'):         # forward and backward of one stage in TWO captures sharing the pool
        what = m.backbone.layer3 if 'layer' in case else None
        x = torch.randn(2, 16, 24, 512, device=dev).to(torch.bfloat16).requires_grad_()
        img = torch.randn(2, 3, 128, 192, device=dev)
        mods = (what,) if what is not None else (m.backbone, m.neck)
        params = [p for mm in mods for p in mm.parameters() if p.requires_grad]
        fwd = (lambda: (what.forward_nhwc(x),)) if what is not None else (lambda: m.extract_feat_nhwc(img))
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                A.grad_arena.new_step()
                o = fwd()
                torch.autograd.grad(o, params, [torch.ones_like(t) for t in o], allow_unused=True)
                A.join_side_streams()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            outs = fwd()
        print('forward captured', flush=True)
        gouts = [torch.ones_like(t) for t in outs]
        with torch.cuda.graph(g2, pool=g.pool(), stream=s):
            A.grad_arena.new_step()
            r = torch.autograd.grad(outs, params, gouts, allow_unused=True)
            if 'join' in case:
                A.join_side_streams()
        print('backward captured', flush=True)
        g.replay(); g2.replay(); g.replay(); g2.replay()
        torch.cuda.synchronize()
    elif case.startswith('gt'):
        from brcnn.graphs import GraphedTrunk
        img = torch.randn(2, 3, 128, 192, device=dev)
        gt = GraphedTrunk(m)
        params = [p for p in gt._params() if p.requires_grad]
        if 'eager' in case:
            import contextlib
            es = torch.cuda.Stream()
            with (torch.cuda.stream(es) if 'estream' in case else contextlib.nullcontext()):
                A.grad_arena.new_step()
                feats = m.extract_feat_nhwc(img)
                if 'fwdonly' not in case:
                    r = torch.autograd.grad(feats, params, [torch.ones_like(f) for f in feats], allow_unused=True)
                    A.join_side_streams()
            torch.cuda.synchronize()
            if 'drop' in case:
                del feats, r
        if 'aten' in case:          # an unrelated eager autograd pass on the default stream beforehand
            w_ = torch.randn(64, 64, device=dev, requires_grad=True)
            torch.autograd.grad((w_ @ w_).sum(), [w_])
            torch.cuda.synchronize()
        gt.seen[gt._key(img)] = 5
        out = gt(img)
        torch.autograd.backward(out, [torch.ones_like(o) for o in out])
        torch.cuda.synchronize()
print('CASE_OK', case, flush=True)
