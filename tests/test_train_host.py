"""Host logic of the training step (vfloodnet_amd.train) that needs no GPU: the scheduler arithmetic, the refusals."""
import pytest
import torch


class _Opt:
    lr = 1e-5


def test_step_lr_matches_torch():
    from vfloodnet_amd.train import StepLR
    o = _Opt()
    s = StepLR(o, step_size=25, gamma=0.5, last_epoch=-1)
    p = torch.nn.Parameter(torch.zeros(1))
    ref_opt = torch.optim.SGD([p], lr=1e-5)
    ref = torch.optim.lr_scheduler.StepLR(ref_opt, step_size=25, gamma=0.5, last_epoch=-1)
    for epoch in range(80):
        assert s.get_last_lr()[0] == pytest.approx(ref.get_last_lr()[0], rel=1e-12), epoch
        ref_opt.step(); ref.step(); s.step()
    # resume (train_video_seg.py:146-147: last_epoch = start_epoch - 1)
    o2 = _Opt()
    s2 = StepLR(o2, step_size=25, gamma=0.5, last_epoch=59)
    assert s2.get_last_lr()[0] == pytest.approx(1e-5 * 0.5 ** 2)


def test_no_cpu_fallback_in_training():
    from vfloodnet_amd import AFB_URR, train as T
    model = AFB_URR(torch.device('cpu'), update_bank=False, _allow_cpu_container=True)
    with pytest.raises(RuntimeError):
        T.AdamW(model.named_parameters(), 1e-5)                       # parameters on the CPU
    with pytest.raises(ValueError):
        T.AdamW([], 1e-5)
    model.train()
    frames, masks = torch.zeros(1, 3, 32, 32), torch.zeros(1, 2, 32, 32)
    with pytest.raises(ValueError):
        T.forward_backward(model, frames, masks)                      # a sample needs at least two frames
    with pytest.raises(RuntimeError):
        T.forward_backward(model, torch.zeros(2, 3, 32, 32), torch.zeros(2, 1, 32, 32))      # single object: as the reference's top-2
    model.eval()
    with pytest.raises(RuntimeError):
        T.forward_backward(model, torch.zeros(2, 3, 32, 32), torch.zeros(2, 2, 32, 32))      # needs model.train()


def test_host_single_threaded_restores_the_setting():
    from vfloodnet_amd.train import _host_single_threaded
    n = torch.get_num_threads()
    with _host_single_threaded():
        assert torch.get_num_threads() == 1
        with _host_single_threaded():
            assert torch.get_num_threads() == 1
    assert torch.get_num_threads() == n
