"""HIP detector ops (through the C ABI) on published / analytic known answers - NOT through oracle/detops_ref.py.
Fixture: tests/golden/detops_known_answers.{json,npz} (generator: oracle/gen_golden_detops.py; detectron2's own ROIAlign unit
test table, identities of the DeformConv definition, the NMS tie rule, the apply_deltas clamp, an exact integer GEMM).
The inputs are small dyadic rationals, so float32 kernels must reproduce the expectations EXACTLY unless a tolerance is stated."""
import numpy as np
import pytest
import torch

from test_known_answers import linear_maps, load

pytestmark = pytest.mark.gpu


def _cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


def test_roi_align_detectron2_unit_test_table(golden_dir):
    from waymo_2d_tracking_amd.detnet.nn import ops
    K, _ = load(golden_dir)
    c = K['roi_align']['detectron2_5x5']
    # the kernel handles channels in runs of 4: replicate the single channel, every copy must carry the table
    img = torch.tensor(c['input'], dtype=torch.float32).view(1, 1, 5, 5).expand(1, 8, 5, 5)
    for roi, key in ((c['roi'], 'aligned_true'), (c['roi_for_aligned_false_on_aligned_kernel'], 'aligned_false')):
        rois = torch.tensor([[0.0] + [float(v) for v in roi]]).cuda()
        got = ops.roi_pool_fpn([_cl(img)], rois, [1.0], pooled=c['output_size'])
        for ch in range(8):
            assert np.array_equal(got[0, ch].cpu().numpy(), np.array(c[key], np.float32)), key


def test_roi_align_linear_field_levels_and_border(golden_dir):
    from waymo_2d_tracking_amd.detnet.nn import ops
    K, _ = load(golden_dir)
    c = K['roi_align']['linear_field']
    maps = linear_maps(c)
    maps = torch.cat([maps, maps[:, :2]], 1)                                  # 8 channels
    rois = torch.tensor([[0.0] + r for r in c['rois']]).cuda()
    got = ops.roi_pool_fpn([_cl(maps)], rois, [c['scale']], pooled=c['pooled'])
    exp = np.array(c['expected'])
    np.testing.assert_allclose(got[:, :6].cpu().double().numpy(), exp, rtol=0, atol=c['tolerance'])
    np.testing.assert_allclose(got[:, 6:].cpu().double().numpy(), exp[:, :2], rtol=0, atol=c['tolerance'])
    c = K['roi_align']['level_assignment']
    feats = [_cl(torch.full((1, 8, 2304 // s, 2304 // s), float(l))) for l, s in zip((2, 3, 4, 5), (4, 8, 16, 32))]
    rois = torch.tensor([[0.0, 64.0, 64.0, 64.0 + s, 64.0 + s] for s in c['box_sizes']]).cuda()       # squares inside the image
    got = ops.roi_pool_fpn(feats, rois, [1 / 4, 1 / 8, 1 / 16, 1 / 32])
    # a constant map pools to its constant: the value tells which level the assignment rule selected
    # (float32 mean over up to 9 x 9 samples whose bilinear weights sum to 1 within an ulp: 1e-5)
    np.testing.assert_allclose(got[:, 0, 3, 3].cpu().numpy(), np.array(c['expected_levels'], np.float32), rtol=0, atol=1e-5)
    np.testing.assert_allclose(got[:, 5, 0, 6].cpu().numpy(), np.array(c['expected_levels'], np.float32), rtol=0, atol=1e-5)
    c = K['roi_align']['border_fraction']
    got = ops.roi_pool_fpn([_cl(torch.ones((1, 8, c['H'], c['W'])))], torch.tensor([[0.0] + c['roi']]).cuda(), [1.0], pooled=c['pooled'])
    np.testing.assert_allclose(got[0, 3].cpu().double().numpy(), np.array(c['expected']), rtol=0, atol=1e-6)


def test_deform_conv_identities_exact(golden_dir):
    """Integer offsets == integer gather + grouped conv; (+.5, +.5) == conv of the 2x2 box-filtered input; offsets beyond the
    border == 0; zero offsets == F.conv2d - for 16 / 32 / 64 channels per group, stride 1 and 2, near and far samples, also
    through the persistent kernel's table pre-pass and with the far-offset hint."""
    from waymo_2d_tracking_amd.detnet.nn import ops
    K, A = load(golden_dir)
    for c in K['deform_conv']['cases']:
        n = c['name']
        x, off, w = (torch.from_numpy(A[n + k]) for k in ('__x', '__offset', '__weight'))
        packed = ops.deform_pack_weight(w.cuda(), c['groups'])
        exp = A[n + '__expected']
        for far in (False, True):
            got = ops.deform_conv3x3(_cl(x), _cl(off), packed, c['groups'], c['stride'], 1, far_offsets=far)
            assert np.array_equal(got.cpu().numpy(), exp), (n, far)
        if c['kind'] == 'zero':
            got = ops.deform_conv3x3(_cl(x), None, packed, c['groups'], c['stride'], 1)       # plain grouped conv path
            assert np.array_equal(got.cpu().numpy(), exp), n


def test_nms_tie_rule_and_deltas_clamp(golden_dir):
    from waymo_2d_tracking_amd.detnet.nn import ops
    K, _ = load(golden_dir)
    c = K['small']['nms_tie']
    b = torch.tensor(c['boxes'], dtype=torch.float32).cuda()
    assert ops.nms_sorted(b, None, c['thr']).cpu().tolist() == c['keep']               # IoU == thr is NOT suppressed
    assert ops.nms_sorted(b, None, 0.49).cpu().tolist() == c['keep_at_0p49']
    assert ops.nms_sorted(b, torch.tensor(c['groups'], dtype=torch.int32).cuda(), c['thr']).cpu().tolist() == c['keep_with_groups']
    c = K['small']['apply_deltas_clamp']
    got = ops.decode_boxes(torch.tensor(c['unit_weight_deltas']).cuda(), torch.tensor(c['unit_weight_boxes'], dtype=torch.float32).cuda(),
                           (1.0, 1.0, 1.0, 1.0))
    np.testing.assert_allclose(got.cpu().numpy(), np.array(c['unit_weight_expected']), rtol=c['rel_tol'], atol=1e-3)


def test_fc_gemm_exact_on_integers(golden_dir):
    from waymo_2d_tracking_amd.detnet.nn import ops
    _, A = load(golden_dir)
    got = ops.gemm_nt(torch.from_numpy(A['fc__a']).cuda(), torch.from_numpy(A['fc__b']).cuda(), torch.from_numpy(A['fc__bias']).cuda(),
                      relu=True)
    assert np.array_equal(got.cpu().numpy(), A['fc__expected'])
