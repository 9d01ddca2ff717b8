"""Command-line flags of the reference's DG_VAE/config.py:4-29, kept verbatim, plus what a box without
the authors' dataset directory needs (--synthetic, --data_dir) and the flags the reference's README
passes but its argparse rejects (--gpus)."""
import argparse


def get_parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Pytorch training script of DG_VAE.')
    parser.add_argument('--exp_id', type=str, default='default', help='Experiment ID')
    parser.add_argument('--local-rank', type=int, default=0, help='Local rank for distributed training')
    # Model
    parser.add_argument('--model', type=str, default='DG_VAE', help='Model name', choices=['DG_VAE', 'DG_AE', 'AE'])
    parser.add_argument('--dim_hidden', type=int, default=64, help='Dimension of hidden layer')
    parser.add_argument('--dim_feature', type=int, default=6, help='Dimension of input feature')
    parser.add_argument('--s_rounds', type=int, default=4, help='Number of rounds for source node')
    parser.add_argument('--t_rounds', type=int, default=4, help='Number of rounds for target node')
    parser.add_argument('--layernorm', action='store_true', help='Enable layernorm')
    # Training
    parser.add_argument('--gpus', default='0', help='accepted for compatibility with the README commands; ignored')
    parser.add_argument('--type', type=str, required=True, choices=['aig', 'mig', 'xmg', 'xag'], help='Circuit type to train')
    parser.add_argument('--batch_size', type=int, default=4, help='Batch size')
    parser.add_argument('--num_epochs', type=int, default=60, help='Number of epochs')
    parser.add_argument('--lr', type=float, default=1e-3, help='Learning rate')
    parser.add_argument('--distributed', action='store_true', help='Enable distributed training')
    parser.add_argument('--resume', action='store_true')
    # additions
    parser.add_argument('--synthetic', type=int, default=0, metavar='N',
                        help='train on N synthetic levelised graphs of --type instead of the npz dataset')
    parser.add_argument('--synthetic_nodes', type=int, default=1024, help='nodes per synthetic graph')
    parser.add_argument('--synthetic_levels', type=int, default=30, help='logic levels per synthetic graph')
    parser.add_argument('--data_dir', type=str, default='', help='directory with graphs.npz (and labels.npz for mig/xag/xmg): '
                        'the reference hard-codes it in train.py')
    parser.add_argument('--device_levels', action='store_true', help='skip the host levelisation of the dataset: every batch is levelised '
                        'on the device when its plan is built (csrc/plan_build.hip)')
    parser.add_argument('--circuit_file', type=str, default='graphs.npz')
    parser.add_argument('--label_file', type=str, default='labels.npz')
    parser.add_argument('--stage_epochs', type=int, nargs=3, default=[100, 60, 60],
                        help='epochs of the three training stages (train.py:81-85)')
    parser.add_argument('--save_dir', type=str, default='./exp')
    return parser.parse_args(argv)
