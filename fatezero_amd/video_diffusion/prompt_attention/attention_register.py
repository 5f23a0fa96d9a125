"""Connect a controller to the UNet (reference: video_diffusion/prompt_attention/attention_register.py:14-258).

The reference monkey-patches `.forward` of every `CrossAttention` / `SparseCausalAttention` module under
`unet.{down,mid,up}*` (skipping children named `attn_temporal`) with a closure that calls
`controller(P, is_cross, place)` between softmax and P.V.  Here the attention modules already execute through the
fused HIP kernels and only need to know *which* controller to ask for a plan, so registration walks the same
module tree with the same rules, stores `(controller, place_in_unet)` on each module, counts the layers (32 for
SD-1.x) and sets `controller.num_att_layers` -- re-registration swaps / detaches controllers like the reference.
"""


class DummyController:
    def __call__(self, *args):
        return args[0]

    def __init__(self):
        self.num_att_layers = 0

    def attention_plan(self, is_cross, place, n_frames, clip_len, heads, lq, lk, device):
        from ..models.attention import AttnPlan
        return AttnPlan(n_frames)

    issue_events_first = True

    def issue_signature(self):
        return ("dummy",)


def register_attention_control(model, controller):
    "Connect a model with a controller"
    if controller is None:
        controller = DummyController()

    def register_recr(name, module, count, place_in_unet):
        if module.__class__.__name__ in ("CrossAttention", "SparseCausalAttention"):
            module.controller = controller
            module.place_in_unet = place_in_unet
            return count + 1
        for child_name, child in module.named_children():
            if child_name != "attn_temporal":
                count = register_recr(child_name, child, count, place_in_unet)
        return count

    cross_att_count = 0
    for name, net in model.unet.named_children():
        if "down" in name:
            cross_att_count += register_recr(name, net, 0, "down")
        elif "up" in name:
            cross_att_count += register_recr(name, net, 0, "up")
        elif "mid" in name:
            cross_att_count += register_recr(name, net, 0, "mid")
    controller.num_att_layers = cross_att_count
    return cross_att_count
