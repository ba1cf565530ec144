"""Module operators with the reference's interface (models_clevr/nmn3_modules.py:11-495).

`Modules(image_feat_grid, word_vecs, num_choices)`; every operator is a method
`XModule([input_0[, input_1]], time_idx, batch_idx)` taking `[Nb]` int32 index vectors and
`[Nb,H,W,1]` attention maps and returning `[Nb,H,W,1]` (attention modules) or
`[Nb,num_choices]` (answer modules), exactly the positional order TensorFlow-Fold's td.Record feeds
them in (models_clevr/nmn3_model.py:57-132).  Each call is one `n2nmn_module_forward` through the
C-ABI (the same HIP kernels the batched program executor launches); tensors are torch CUDA tensors.
Trailing keyword arguments of the reference (map_dim, scope, reuse, pos_val, kernel_size) are
accepted and checked against the committed model where they matter.
"""
from __future__ import annotations

from .engine import Engine


class Modules:
    def __init__(self, image_feat_grid, word_vecs, num_choices, engine: Engine = None):
        if engine is None:
            # the reference's positional form (models_clevr/nmn3_modules.py:11, exp_shapes/visualize_shapes.ipynb
            # cell 6): the weights are "the graph's"; here, those of the most recently built root Engine
            engine = Engine.latest()
            if engine is None:
                raise ValueError('Modules(image_feat_grid, word_vecs, num_choices): no Engine has been built yet -- '
                                 'build the model (NMN3Model / Engine) first or pass engine=')
        if num_choices != engine.dims.num_choices:
            raise ValueError('num_choices differs from the committed model')
        self.engine = engine
        self.image_feat_grid = image_feat_grid
        self.word_vecs = word_vecs
        self.num_choices = num_choices
        d = engine.dims
        self.att_shape = [None, d.H, d.W, 1]

    def bind(self, image_feat_grid, word_vecs):
        """Re-point the operators at new activations (the eager analogue of feeding placeholders)."""
        self.image_feat_grid = image_feat_grid
        self.word_vecs = word_vecs
        return self

    def _run(self, name, inputs, time_idx, batch_idx, map_dim=None, kernel_size=None):
        d = self.engine.dims
        if map_dim is not None and map_dim != d.map_dim:
            raise ValueError('map_dim differs from the committed model')
        if kernel_size is not None and kernel_size != d.kernel_size:
            raise ValueError('kernel_size differs from the committed model')
        return self.engine.module_forward(name, inputs, time_idx, batch_idx,
                                          self.image_feat_grid, self.word_vecs)

    # attention modules ----------------------------------------------------------------------
    def SceneModule(self, time_idx, batch_idx, pos_val=3, scope='SceneModule', reuse=True):
        if pos_val != 3:
            raise ValueError('SceneModule is built with pos_val=3 (nmn3_modules.py:60)')
        return self._run('_Scene', [], time_idx, batch_idx)

    def FindModule(self, time_idx, batch_idx, map_dim=250, scope='FindModule', reuse=True):
        return self._run('_Find', [], time_idx, batch_idx, map_dim)

    def FilterModule(self, input_0, time_idx, batch_idx, map_dim=250, scope='FilterModule',
                     reuse=True):
        return self._run('_Filter', [input_0], time_idx, batch_idx, map_dim)

    def FindSamePropertyModule(self, input_0, time_idx, batch_idx, map_dim=250,
                               scope='FindSamePropertyModule', reuse=True):
        return self._run('_FindSameProperty', [input_0], time_idx, batch_idx, map_dim)

    def TransformModule(self, input_0, time_idx, batch_idx, kernel_size=5, map_dim=250,
                        scope='TransformModule', reuse=True):
        return self._run('_Transform', [input_0], time_idx, batch_idx, map_dim, kernel_size)

    def AndModule(self, input_0, input_1, time_idx, batch_idx, scope='AndModule', reuse=True):
        return self._run('_And', [input_0, input_1], time_idx, batch_idx)

    def OrModule(self, input_0, input_1, time_idx, batch_idx, scope='OrModule', reuse=True):
        return self._run('_Or', [input_0, input_1], time_idx, batch_idx)

    # answer modules -------------------------------------------------------------------------
    def ExistModule(self, input_0, time_idx, batch_idx, scope='ExistModule', reuse=True):
        return self._run('_Exist', [input_0], time_idx, batch_idx)

    def CountModule(self, input_0, time_idx, batch_idx, scope='CountModule', reuse=True):
        return self._run('_Count', [input_0], time_idx, batch_idx)

    def EqualNumModule(self, input_0, input_1, time_idx, batch_idx, scope='EqualNumModule',
                       reuse=True):
        return self._run('_EqualNum', [input_0, input_1], time_idx, batch_idx)

    def MoreNumModule(self, input_0, input_1, time_idx, batch_idx, scope='MoreNumModule',
                      reuse=True):
        return self._run('_MoreNum', [input_0, input_1], time_idx, batch_idx)

    def LessNumModule(self, input_0, input_1, time_idx, batch_idx, scope='LessNumModule',
                      reuse=True):
        return self._run('_LessNum', [input_0, input_1], time_idx, batch_idx)

    def SamePropertyModule(self, input_0, input_1, time_idx, batch_idx, map_dim=250,
                           scope='SamePropertyModule', reuse=True):
        return self._run('_SameProperty', [input_0, input_1], time_idx, batch_idx, map_dim)

    def DescribeModule(self, input_0, time_idx, batch_idx, map_dim=250, scope='DescribeModule',
                       reuse=True):
        return self._run('_Describe', [input_0], time_idx, batch_idx, map_dim)
