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
    # resume (train_video_seg.py:129,146-147: optimizer.load_state_dict restores the DECAYED rate, then
    # StepLR(last_epoch = start_epoch - 1)): the schedule continues, it does not decay a second time
    for saved_epoch in (30, 49, 50, 74):
        o1, r1_p = _Opt(), torch.nn.Parameter(torch.zeros(1))
        r1 = torch.optim.SGD([r1_p], lr=1e-5)
        s1, rs1 = StepLR(o1, 25, 0.5), torch.optim.lr_scheduler.StepLR(r1, 25, 0.5)
        for _ in range(saved_epoch):
            r1.step(); rs1.step(); s1.step()
        ref_sd = r1.state_dict()                              # what the checkpoint of epoch `saved_epoch` holds
        assert ref_sd['param_groups'][0]['initial_lr'] == o1.initial_lr == 1e-5
        o2 = _Opt()
        o2.lr, o2.initial_lr = o1.lr, o1.initial_lr          # AdamW.load_state_dict
        r2 = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-5)
        r2.load_state_dict(ref_sd)
        s2, rs2 = StepLR(o2, 25, 0.5, last_epoch=saved_epoch), torch.optim.lr_scheduler.StepLR(r2, 25, 0.5, last_epoch=saved_epoch)
        for epoch in range(saved_epoch + 1, 110):
            assert s2.get_last_lr()[0] == pytest.approx(rs2.get_last_lr()[0], rel=1e-12), (saved_epoch, epoch)
            assert s2.get_last_lr()[0] == pytest.approx(1e-5 * 0.5 ** (epoch // 25), rel=1e-12)
            r2.step(); rs2.step(); s2.step()
    with pytest.raises(KeyError):                             # torch refuses a resumed optimizer without 'initial_lr'
        StepLR(_Opt(), 25, 0.5, last_epoch=10)


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
