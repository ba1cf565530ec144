"""`from tensorflow.python.ops.nn import dropout` (models_vqa/question_prior_net.py:6)."""
import tensorflow as _tf

dropout = _tf.nn.dropout
