"""Key / value content of three of the reference's training YAMLs (configs/base_fno.yaml, minchan_rno.yaml, matlab_rno.yaml:
settings data, comments dropped, duplicate keys kept where the files repeat them) for the run-plan tests of
pde_policylearning_amd.train_observer: the loop must consume these files unchanged (BASELINE.json north_star)."""

BASE_FNO = """
DATA_FOLDER: './data/planes_channel180_minchan'
project_name: 'fno_vs_unet'
exp_name: '31-FNO-reproduce'
path_name: planes_channel180_minchan
model_name: FNO2dObserver
learning_rate: 0.001
weight_decay: 0.0001
epochs: 500
step_size: 100
gamma: 0.5
modes: 12
width: 32
close_wandb: false
batch_size: 20
downsample_rate: 1
x_range: 32
y_range: 32
ntrain: 7500
ntest: 2500
use_v_plane: false
use_patch: false
timestep: -1
recurrent_model: false
random_split: true
"""

MINCHAN_RNO = """
DATA_FOLDER: './data/planes_channel180_minchan'
ntrain: 8000
ntest: 2000
project_name: 'fno_vs_unet'
exp_name: '28-RNO-reproduce'
path_name: planes_channel180_minchan
model_name: RNO2dObserver
learning_rate: 0.001
weight_decay: 0.0001
step_size: 100
gamma: 0.5
modes: 12
width: 32
downsample_rate: 1
x_range: 32
y_range: 32
use_v_plane: false
use_patch: false
timestep: 2
recurrent_model: true
recurrent_index: 0
random_split: false
width: 34
batch_size: 32
layer_num: 3
close_wandb: true
epochs: 200
"""

MATLAB_RNO = """
DATA_FOLDER: './outputs/0077-add-dudt'
ntrain: 280
ntest: 20
project_name: 'control_v2'
exp_name: '0077-add-dudt'
path_name: planes_channel180_minchan
run_control: true
display_variables:
  - exp_name
  - policy_name
  - pde_loss_weight
  - model_timestep
env_name: NSControlEnvMatlab
model_name: PINObserverFullField
dataset_name: FullFieldNSDataset
init_cond_path: ./data/channel180_minchan_mf.mat
vis_sample_img: false
noise_scale: 0.0
detect_plane: 24
test_plane: -25
w_weight: 0.0
x_range: 32
y_range: 32
fix_flow: true
Re: -1
bc_type: original
policy_name:
  - optimal-observer
rand_scale: 1
reward_type: mse
collect_data: true
collect_start: 0
full_field: true
dump_state: false
pde_loss_weight: 1.0
use_spectral_conv: false
learning_rate: 0.001
weight_decay: 0.0001
step_size: 100
gamma: 0.5
modes: 12
width: 32
downsample_rate: 1
use_v_plane: false
use_patch: false
model_timestep: 1
recurrent_model: true
recurrent_index: 0
random_split: false
plane_indexs: [-10, -8, -6]
vis_frame: -1
vis_interval: -1
show_spatial_dist_interval: 50
output_dir: ./outputs
width: 34
batch_size: 32
layer_num: 1
close_wandb: true
epochs: 100
control_timestep: 2000
"""
