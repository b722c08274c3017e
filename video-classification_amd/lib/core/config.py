"""Global configuration: same keys and defaults as the reference's edict
(/root/reference/lib/core/config.py:12-69) on a dependency-free attribute dict
(easydict is not required).  Added keys: MODEL.CLIP_LEN, MODEL.CROP_SIZE, MODEL.DTYPE,
TRAIN.BUCKET_MB (gradient all-reduce bucket size), TRAIN.GRAD_DTYPE, TEST.NUM_CLIPS / NUM_CROPS.
"""


class AttrDict(dict):
    """dict with attribute access, nested like easydict's EasyDict."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        out = AttrDict()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, AttrDict) else (list(v) if isinstance(v, list) else v)
        return out


config = AttrDict()

# common
config.GPUS = '0'
config.WORKERS = 0
config.DATASET = 'FCVID'
config.OUTPUT_DIR = 'experiments'

# model
config.MODEL = AttrDict()
config.MODEL.LSTM_INDIM = 2048
config.MODEL.LSTM_OUTDIM = 2048
config.MODEL.CLFDIM = 100
config.MODEL.RESNET_TYPE = 18
config.MODEL.INIT_WEIGHTS = True
config.MODEL.PRETRAINED_PATH = 'pretrained_models/?'
config.MODEL.CLIP_LEN = 16
config.MODEL.CROP_SIZE = 112
config.MODEL.DTYPE = 'bf16'
# the reference's recurrent head (lstm.py, model.py:18-25): backbone features -> LSTMCell -> action / class / critic heads.
# Off by default: north_star's hot path is the 3D backbone with a plain classifier head.
config.MODEL.USE_LSTM = False
# raw-frame input (uint8 N x Ts x Hs x Ws x 3 from a decoder): sampling / normalisation done on the GPU by
# utils.ChooseInput -> vc_clip_pack_u8; mean / std in units of u8 / 255
config.MODEL.FRAME_STEP = 1
config.MODEL.MEAN = [0.43216, 0.394666, 0.37645]
config.MODEL.STD = [0.22803, 0.22145, 0.216989]

# training
config.TRAIN = AttrDict()
config.TRAIN.LR = 0.001
config.TRAIN.LR_DECAY_RATE = 1
config.TRAIN.LR_MILESTONES = []
config.TRAIN.OPTIMIZER = 'adam'
config.TRAIN.MOMENTUM = 0.9
config.TRAIN.WD = 0.0001
config.TRAIN.NESTEROV = False
config.TRAIN.R_ACT = 1
config.TRAIN.R_CRT = 1
config.TRAIN.TRAIN_STEP = 10
config.TRAIN.ROLLOUT_STEP = 1
config.TRAIN.GAMMA = 0.99  # discount of the one-step advantage in utils.rollout
config.TRAIN.BEGIN_EPOCH = 0
config.TRAIN.END_EPOCH = 100
config.TRAIN.BATCH_SIZE = 32
config.TRAIN.SHUFFLE = True
config.TRAIN.PRINT_EVERY = 100
config.TRAIN.BUCKET_MB = 32
config.TRAIN.GRAD_DTYPE = ''  # wire format of the gradient all-reduce: '' / 'fp32' (default, as DataParallel sums) or 'bf16'
config.TRAIN.RESUME = ''  # checkpoint.pth written by train.py every epoch: model + optimizer + scheduler + epoch

# testing
config.TEST = AttrDict()
config.TEST.TEST_STEP = 10
config.TEST.BATCH_SIZE = 32
config.TEST.PRINT_EVERY = 100
config.TEST.NUM_CLIPS = 10
config.TEST.NUM_CROPS = 3
