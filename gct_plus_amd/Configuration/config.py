"""Command-line flags of train1.py -- names, types and defaults are the reference's
(Configuration/config.py:1-64: single-dash long flags), because the Bashscript/ drivers pass
them verbatim.  Additions of this build are grouped at the end and are all optional."""


def model_opts(parser):
    parser.add_argument('-N', type=int, default=6, help="# of encoder/decoder")
    parser.add_argument('-H', type=int, default=8, help="heads of attention")
    parser.add_argument('-d_ff', type=int, default=2048)
    parser.add_argument('-d_model', type=int, default=512)
    parser.add_argument('-latent_dim', type=int, default=128)
    parser.add_argument('-dropout', type=float, default=0.1)
    parser.add_argument('-variational', type=bool, default=True)   # truthy for any string, as upstream
    parser.add_argument('-use_cond2dec', action='store_true')
    parser.add_argument('-use_cond2lat', action='store_true')
    parser.add_argument('-get_attn', action='store_true')


def train_opts(parser):
    model_opts(parser)
    parser.add_argument('-seed', type=int)
    parser.add_argument('-start_epoch', type=int, default=1)
    parser.add_argument('-num_epoch', type=int, default=30)
    parser.add_argument('-batch_size', type=int, default=32)
    parser.add_argument('-property_list', nargs='+', default=[])
    parser.add_argument('-model_type', type=str, required=True)
    parser.add_argument('-model_folder', type=str, required=True)
    parser.add_argument('-use_scaffold', action='store_true')
    parser.add_argument('-randomize_prob', type=float, default=0)
    parser.add_argument('-train_params', type=str, nargs='+')
    parser.add_argument('-prepared_folder', type=str, default='./Data/prepared')
    parser.add_argument('-util_folder', type=str, default='./Data/utils')
    parser.add_argument('-debug', action='store_true')
    # KL annealing
    parser.add_argument('-use_KLA', type=bool, default=True)
    parser.add_argument('-KLA_ini_beta', type=float, default=0.02)
    parser.add_argument('-KLA_inc_beta', type=float, default=0.02)
    parser.add_argument('-KLA_max_beta', type=float, default=1.0)
    parser.add_argument('-KLA_beg_epoch', type=int, default=1)
    # learning-rate schedule / Adam
    parser.add_argument('-lr_scheduler', type=str, default="WarmUpDefault")
    parser.add_argument('-lr_WarmUpSteps', type=int, default=8000)
    parser.add_argument('-lr', type=float, default=0.0001)
    parser.add_argument('-lr_beta1', type=float, default=0.9)
    parser.add_argument('-lr_beta2', type=float, default=0.98)
    parser.add_argument('-lr_eps', type=float, default=1e-9)
    # ---- additions of this build (not in the reference) ----
    parser.add_argument('-synthetic', type=int, default=0,
                        help="train on N synthetic MOSES-shaped samples instead of Data/prepared")
    parser.add_argument('-synthetic_valid', type=int, default=0)
    parser.add_argument('-max_strlen', type=int, default=80)
    parser.add_argument('-print_every', type=int, default=1)
    parser.add_argument('-eps_mode', type=str, default='device', choices=['device', 'cpu'],
                        help="'cpu' draws the VAE eps from the CPU torch generator (parity runs)")
