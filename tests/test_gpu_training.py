"""One training step on the device against the CPU restatement of GenRelMSE + CPUBackProp
(DenoiserNN.cpp:959-1103, 1461-1474).

The reference adds learningRate * G * input to the weights pixel by pixel in fp32; the
device reduces the 56 sums of G * input in double and adds once.  G itself is a product of
seven Gaussians divided by a cubed bandwidth: one-ulp differences in exp() move it by ~1e-6
relative.  Tolerances: relMSE 1e-5 relative; each weight increment within 1e-4 of the
largest increment of its layer plus 4e-6 absolute (the restatement rounds to the weight's own
ulp, up to 4.8e-7, at every one of its ~10^4 additions)."""
import numpy as np
import pytest

from conftest import PIN_SEED, scene_path, weights_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,nwin", [(48, 27, 1), (40, 40, 2), (64, 36, 3),
                                      (5, 3, 1), (17, 16, 2), (129, 17, 1), (33, 9, 5), (1, 1, 1)])
def test_train_step_matches_cpu_backprop(oracle, W, H, nwin):
    import ptk
    O = oracle
    spp = 8
    sc = O.load_scene(scene_path("MultiLightGlass"))
    consts = O.build_constants(sc, W, H, 4, 0.005, True)
    ctx = ptk.Context(0)
    ctx.set_constants(consts)
    with pytest.raises(ptk.PtkError):
        ctx.train_step(O.load_weights(weights_path("BEST")))     # no target yet
    # target: a longer render; then the noisy frame that is denoised
    ctx.render(64, ptk.SEED_COUNTER)
    ctx.save_target()
    target = ctx.read_planes(0, 3).T / 64.0
    ctx.reset()
    ctx.render(spp, ptk.SEED_COUNTER)
    acc = ctx.read_planes()
    w = O.load_weights(weights_path("box_grid"))
    lr = 1e-3   # large enough that the increments stand clear of the weights' own ulps

    (n12, n23, n34), rel = ctx.train_step(w, nwin, lr)
    assert ctx.timing().train_ms > 0.0

    feats, var, l2, l3 = O.forward_prop_layers(acc, W, H, spp, w, threads=4)
    den, wc = O.filter(acc, var, W, H, spp, nwin)
    rel_ref = O.rel_mse(den, target, W, H, spp)
    assert abs(rel - rel_ref) <= 1e-5 * abs(rel_ref) + 1e-6
    r12, r23, r34 = O.back_prop(acc, den, wc, var, feats, l2, l3, target, W, H, spp, nwin, lr, w)
    for got, ref, old in ((n12, r12, w[0]), (n23, r23, w[1]), (n34, r34, w[2])):
        d_got, d_ref = got - old, ref - old
        scale = np.abs(d_ref).max()
        assert scale > 0 or W * H == 1        # a single pixel has no neighbour to differ from
        assert np.abs(d_got - d_ref).max() <= 1e-4 * scale + 4e-6
        # increments repeat with the layer's fan-in, as the reference's update does
    assert np.allclose((n12 - w[0])[:36], (n12 - w[0])[36:72], rtol=0, atol=1e-6)

    # a host-supplied target is the same as the saved one
    ctx.set_target(target.astype(np.float32))
    (m12, _, _), rel2 = ctx.train_step(w, nwin, lr)
    assert np.allclose(m12, n12, rtol=0, atol=1e-7) and abs(rel2 - rel) <= 1e-6 * abs(rel)
    ctx.close()


def test_train_step_refuses_portrait_images(oracle):
    """The reference's filter derivative reads outside the image when height > width."""
    import ptk
    O = oracle
    sc = O.load_scene(scene_path("cornell"))
    ctx = ptk.Context(0)
    ctx.set_constants(O.build_constants(sc, 27, 48, 4, 0.005, True))
    ctx.render(4, ptk.SEED_COUNTER)
    ctx.save_target()
    with pytest.raises(ptk.PtkError):
        ctx.train_step(O.load_weights(weights_path("BEST")))
    ctx.close()
