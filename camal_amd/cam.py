"""``CAM`` — mirror of ``src/helpers/class_activation_map.py:24-74``.

``CAM(model, device, last_conv_layer, fc_layer_name).run(instance, returned_cam_for_label=1)``
returns ``(cam[L] float32, label_pred int, proba[2] float32)`` exactly like the reference,
but the forward, the "hook" on the last block and the ``W_fc[label]·F`` product all happen
inside one ``camal_forward_cam`` call: the 128 x L feature map never leaves the GPU (it is
never even written to HBM).  ``last_conv_layer = model._modules["layers"][1]`` hooks the second block instead
(``camal_forward_cam_tap``: a channel-dot kernel on that block's output).  ``run_batch`` is the batched form ``CamAL.predict`` needs.
"""
from __future__ import annotations

import numpy as np


class CAM:
    def __init__(self, model, device, last_conv_layer="layer3", fc_layer_name="fc1", verbose=False):
        self.device = device
        self.last_conv_layer = last_conv_layer
        self.fc_layer_name = fc_layer_name
        self.model = model
        self.verbose = verbose
        # the hooked layer: a ResNetBlock handle (model._modules["layers"][i]); the default name strings of the reference's
        # signature mean the last block.  Block 0's output is narrower than the head's weights: the reference's
        # W_fc[label].dot(features) fails on it (class_activation_map.py:45), so does this.
        self.tap_block = getattr(last_conv_layer, "index", 2)
        if self.tap_block not in (1, 2):
            raise ValueError("last_conv_layer must be the second or the last ResNetBlock: the first one's output has "
                             "fewer channels than the linear layer's weights (shapes not aligned)")

    def run(self, instance, label_instance=None, returned_cam_for_label=None):
        assert label_instance is not None or returned_cam_for_label is not None
        cams, labels, probs = self.run_batch(np.array(instance)[None, :], returned_cam_for_label)
        if self.verbose:
            print("Pred proba :", probs[0])
        return cams[0], int(labels[0]), probs[0]

    def run_batch(self, data, returned_cam_for_label=1):
        """data [N,L] -> (cams [N,L] f32, labels [N] int, probs [N,2] f32) as NumPy arrays."""
        eng = self.model.engine()
        x = np.asarray(data)
        want = 1 if returned_cam_for_label is None else int(returned_cam_for_label)
        logits, cam = eng.forward_cam(x, cam_label=want, tap_block=self.tap_block)
        lg = logits[0].cpu().numpy()
        cams = cam[0].cpu().numpy()
        # F.softmax(logits, dim=1) and topk(1) (class_activation_map.py:59,64) on two float32 logits per window: host
        e = np.exp(lg - lg.max(axis=1, keepdims=True))
        probs = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
        labels = (probs[:, 1] > probs[:, 0]).astype(np.int32)  # a tie goes to class 0
        if returned_cam_for_label is None:
            # class_activation_map.py:69-70: CAM of the predicted class; redo the rows that predicted 0
            zero = np.flatnonzero(labels == 0)
            if len(zero):
                _, cam0 = eng.forward_cam(x[zero], cam_label=0, tap_block=self.tap_block)
                cams[zero] = cam0[0].cpu().numpy()
        return cams, labels, probs
