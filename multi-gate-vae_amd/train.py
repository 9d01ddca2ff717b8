"""Entry point with the shape of the reference's DG_VAE/train.py:21-109: build encoder + per-type Model,
Trainer, then the three-stage loss-weight schedule [1,0,0] -> [1,5,0] -> [1,4,4] at lr 1e-4, lr_step 50.

    python train.py --exp_id e --model DG_AE --type aig --layernorm --batch_size 4 --synthetic 32
    torchrun --nproc_per_node=8 --master-addr 127.0.0.1 train.py ... --distributed

The reference hard-codes the authors' dataset directory and `distributed=True`; here `--distributed`
is honoured, `--data_dir` points at the MixGate npz files (deepgate.NpzParser) and `--synthetic N` generates N
levelised DAGs instead."""
import os

import deepgate
import deepgate.dg_ae_model_aig
import deepgate.dg_ae_model_mig
import deepgate.dg_ae_model_xag
import deepgate.dg_ae_model_xmg
import deepgate.digae_layer
from config import get_parse_args
from deepgate import synthetic


def main(argv=None):
    args = get_parse_args(argv)
    model_map = {
        'aig': deepgate.dg_ae_model_aig.Model, 'mig': deepgate.dg_ae_model_mig.Model,
        'xmg': deepgate.dg_ae_model_xmg.Model, 'xag': deepgate.dg_ae_model_xag.Model,
    }
    print('[INFO] Parse Dataset')
    if args.synthetic > 0:
        n_in = max(args.synthetic_nodes // 16, 1)
        graphs = [synthetic.make_graph(args.type, args.synthetic_nodes, args.synthetic_levels, 100 + i, n_inputs=n_in)
                  for i in range(args.synthetic)]
        cut = max(int(len(graphs) * 0.9), 1)
        train_dataset, val_dataset = graphs[:cut], graphs[cut:]
    elif args.data_dir:
        dataset = deepgate.NpzParser(args.data_dir, os.path.join(args.data_dir, args.circuit_file),
                                     os.path.join(args.data_dir, args.label_file), args.type, levelise=not args.device_levels)
        train_dataset, val_dataset = dataset.get_dataset()
    else:
        raise SystemExit('pass --data_dir DIR (graphs.npz [+ labels.npz]) or --synthetic N')

    print('[INFO] Create Model')
    if 'DG' not in args.model:
        raise SystemExit('--model AE (DirectedGCNConvEncoder) is outside the accelerated path')
    encoder = deepgate.digae_layer.DirectMultiGCNEncoder(
        dim_hidden=args.dim_hidden, dim_feature=args.dim_feature, enable_reverse=True,
        s_rounds=args.s_rounds, t_rounds=args.t_rounds, layernorm=args.layernorm)
    if 'VAE' in args.model:
        raise SystemExit('the DG_VAE training path does not run in the reference either (SURVEY.md §3.4); '
                         'the sampler/KL operators are available as deepgate.digvae_model.DirectedGVAE')
    model = model_map[args.type](struct_encoder=encoder, dim_hidden=args.dim_hidden, enable_encode=True, enable_reverse=True)
    trainer = deepgate.Trainer(args, model, training_id=args.exp_id, save_dir=args.save_dir, batch_size=args.batch_size,
                               device='cuda:0', distributed=args.distributed)
    if args.resume:
        trainer.resume()
    stage_configs = [
        {'epochs': args.stage_epochs[0], 'weights': [1.0, 0.0, 0.0], 'lr': 1e-4},
        {'epochs': args.stage_epochs[1], 'weights': [1.0, 5.0, 0.0], 'lr': 1e-4},
        {'epochs': args.stage_epochs[2], 'weights': [1.0, 4.0, 4.0], 'lr': 1e-4},
    ]
    for stage_idx, config in enumerate(stage_configs):
        print('\n' + '=' * 40)
        print('[STAGE %d] Start Training' % (stage_idx + 1))
        print('|-- Epochs: %d\n|-- Loss Weights: %s\n|-- Learning Rate: %g' % (config['epochs'], config['weights'], config['lr']))
        trainer.set_training_args(rc_prob_func_weight=config['weights'], lr=config['lr'], lr_step=50)
        trainer.train(config['epochs'], train_dataset, val_dataset)
        if trainer.rank == 0:
            trainer.save(os.path.join(trainer.log_dir, 'stage_%d.pth' % (stage_idx + 1)))
    print('\n[INFO] All training stages completed!')


if __name__ == '__main__':
    main()
