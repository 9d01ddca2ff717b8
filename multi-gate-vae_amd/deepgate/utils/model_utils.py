"""Checkpoint loading with the reference's conventions (utils/model_utils.py:3-66): strip a DataParallel
`module.` prefix, keep own values for missing or shape-mismatched entries, optionally resume Adam."""
import torch


def load_model(model, model_path, optimizer=None, local_rank=0, device='cuda'):
    checkpoint = torch.load(model_path, map_location='cpu')
    if local_rank == 0:
        print('loaded {}, epoch {}'.format(model_path, checkpoint['epoch']))
    state = {}
    for k, v in checkpoint['state_dict'].items():
        state[k[7:] if k.startswith('module') and not k.startswith('module_list') else k] = v
    own = model.state_dict()
    for k in list(state):
        if k not in own:
            if local_rank == 0:
                print('Drop parameter {}.'.format(k))
        elif state[k].shape != own[k].shape:
            if local_rank == 0:
                print('Skip loading parameter {}, required shape{}, loaded shape{}.'.format(k, own[k].shape, state[k].shape))
            state[k] = own[k]
    for k in own:
        if k not in state:
            if local_rank == 0:
                print('No param {}.'.format(k))
            state[k] = own[k]
    model.load_state_dict(state, strict=False)
    if optimizer is None:
        return model
    start_epoch = 0
    if 'optimizer' in checkpoint:
        optimizer.load_state_dict(checkpoint['optimizer'])
        start_epoch = checkpoint['epoch']
        if local_rank == 0:
            print('Resumed optimizer with epoch', start_epoch)
    elif local_rank == 0:
        print('No optimizer parameters in checkpoint.')
    return model, optimizer, start_epoch
