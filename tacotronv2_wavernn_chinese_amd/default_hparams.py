# Default hyper-parameters of the mel->wav path: the names and values the
# reference reads from wavernn_hparams.py (:18-28, :35-41, :50, :55-57).
sample_rate = 22050
num_mels = 80
hop_length = 275
bits = 10
mu_law = True

voc_model_id = 'wavernn'
voc_mode = 'RAW'
voc_upsample_factors = (5, 5, 11)
voc_rnn_dims = 512
voc_fc_dims = 512
voc_compute_dims = 128
voc_res_out_dims = 128
voc_res_blocks = 10
voc_pad = 2
voc_gen_at_checkpoint = 5
voc_gen_batched = False
voc_target = 11_000
voc_overlap = 550

# training (wavernn_hparams.py:5, :44-52)
feature_path = './wavernn_training_data.txt'
voc_batch_size = 32
voc_lr = 1e-4
voc_checkpoint_every = 1000
voc_total_steps = 500_000
voc_test_samples = 50
voc_seq_len = hop_length * 5
voc_clip_grad_norm = 4
