"""Host-side mirror of the reference's ``model/egtr.py``: ``DetrForSceneGraphGeneration`` and the SGG loss.

Same constructor / forward signature, output object and state-dict keys as the reference.
The relation head (model/egtr.py:322-418) is evaluated through the separable algebra of DESIGN.md: six small
GEMMs produce per-query tables, and ONE fused HIP kernel (egtr_amd/csrc/rel_head.hip) does the pairwise gate,
gated sum, both 3-layer MLPs and the frequency-bias gather -- the reference's 573 MB ``relation_source`` tensor
never exists.  Citations "egtr:NNN" are to /root/reference/model/egtr.py.
"""
import copy
import math
import os
import random
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .deformable_detr import (DeformableDetrHungarianMatcher, DeformableDetrMLPPredictionHead, DeformableDetrModel,
                              MatchedIndices, detection_heads,
                              DeformableDetrPreTrainedModel, inverse_sigmoid)
from .hf_compat import ModelOutput
from .util import (center_to_corners_format, dice_loss, generalized_box_iou, nested_tensor_from_tensor_list,
                   sigmoid_focal_loss)


@dataclass
class DetrSceneGraphGenerationOutput(ModelOutput):
    """egtr:53-115."""
    loss: Optional[torch.FloatTensor] = None
    loss_dict: Optional[Dict] = None
    logits: Optional[torch.FloatTensor] = None
    pred_boxes: Optional[torch.FloatTensor] = None
    pred_rel: Optional[torch.FloatTensor] = None
    pred_connectivity: Optional[torch.FloatTensor] = None
    auxiliary_outputs: Optional[List[Dict]] = None
    last_hidden_state: Optional[torch.FloatTensor] = None
    decoder_hidden_states: Optional[Tuple[torch.FloatTensor]] = None
    decoder_attentions: Optional[Tuple[torch.FloatTensor]] = None
    cross_attentions: Optional[Tuple[torch.FloatTensor]] = None
    encoder_last_hidden_state: Optional[torch.FloatTensor] = None
    encoder_hidden_states: Optional[Tuple[torch.FloatTensor]] = None
    encoder_attentions: Optional[Tuple[torch.FloatTensor]] = None


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


# False: slot projections, first MLP layer and gate logits of the relation head as three launches (the A/B twin of the merged
# preparation below; a module attribute that tests patch -- no environment switch since round 6)
REL_PREP_MERGED = True


class DetrForSceneGraphGeneration(DeformableDetrPreTrainedModel):
    def __init__(self, config, **kwargs):
        super().__init__(config)
        self.model = DeformableDetrModel(config)
        self.class_embed = nn.Linear(config.d_model, config.num_labels)
        self.bbox_embed = DeformableDetrMLPPredictionHead(input_dim=config.d_model, hidden_dim=config.d_model,
                                                          output_dim=4, num_layers=3)
        prior_prob = 0.01
        bias_value = -math.log((1 - prior_prob) / prior_prob)
        self.class_embed.bias.data = torch.ones(config.num_labels) * bias_value
        nn.init.constant_(self.bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(self.bbox_embed.layers[-1].bias.data, 0)
        # two-stage: the last class / box head scores the encoder tokens (region proposals, egtr:142-145)
        num_pred = (config.decoder_layers + 1) if config.two_stage else config.decoder_layers
        if config.with_box_refine:
            self.class_embed = _get_clones(self.class_embed, num_pred)
            self.bbox_embed = _get_clones(self.bbox_embed, num_pred)
            nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            self.model.decoder.bbox_embed = self.bbox_embed
        else:  # every index aliases ONE module (egtr:152-158)
            nn.init.constant_(self.bbox_embed.layers[-1].bias.data[2:], -2.0)
            self.class_embed = nn.ModuleList([self.class_embed for _ in range(num_pred)])
            self.bbox_embed = nn.ModuleList([self.bbox_embed for _ in range(num_pred)])
            self.model.decoder.bbox_embed = None
        if config.two_stage:   # egtr:159-163
            self.model.decoder.class_embed = self.class_embed
            for box_embed in self.bbox_embed:
                nn.init.constant_(box_embed.layers[-1].bias.data[2:], 0.0)

        self.num_queries = self.config.num_queries
        self.head_dim = config.d_model // config.num_attention_heads
        self.layer_head = self.config.decoder_layers * config.num_attention_heads

        fg_matrix = kwargs.get("fg_matrix", None)
        if fg_matrix is not None:  # training: frequency-bias tables from dataset statistics (egtr:169-184)
            eps = config.freq_bias_eps
            rel_dist = torch.FloatTensor((fg_matrix.sum(axis=(0, 1))) / (fg_matrix.sum() + eps))
            # NB the reference's operator precedence: fg + (eps / (sum + eps)), kept bug-for-bug
            triplet_dist = torch.FloatTensor(fg_matrix + eps / (fg_matrix.sum(2, keepdims=True) + eps))
            triplet_dist = F.log_softmax(triplet_dist, dim=-1) if config.use_log_softmax else triplet_dist.log()
            self.rel_dist = nn.Parameter(rel_dist, requires_grad=False)
            self.triplet_dist = nn.Parameter(triplet_dist, requires_grad=False)
        else:  # inference: filled from the checkpoint (egtr:185-194); zero-initialised here rather than garbage
            self.triplet_dist = nn.Parameter(
                torch.zeros(config.num_labels + 1, config.num_labels + 1, config.num_rel_labels), requires_grad=False)
            self.rel_dist = nn.Parameter(torch.ones(config.num_rel_labels), requires_grad=False)

        d = config.d_model
        self.proj_q = nn.ModuleList([nn.Linear(d, d) for _ in range(config.decoder_layers)])
        self.proj_k = nn.ModuleList([nn.Linear(d, d) for _ in range(config.decoder_layers)])
        self.final_sub_proj = nn.Linear(d, d)
        self.final_obj_proj = nn.Linear(d, d)
        self.rel_predictor_gate = nn.Linear(2 * d, 1)
        self.rel_predictor = DeformableDetrMLPPredictionHead(input_dim=2 * d, hidden_dim=d,
                                                             output_dim=config.num_rel_labels, num_layers=3)
        self.connectivity_layer = DeformableDetrMLPPredictionHead(input_dim=2 * d, hidden_dim=d, output_dim=1,
                                                                  num_layers=3)
        self.post_init()

    @torch.jit.unused
    def _set_aux_loss(self, outputs_class, outputs_coord):
        return [{"logits": a, "pred_boxes": b} for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]

    # ---------------------------------------------------------------------------------------- relation head
    def _relation_head(self, queries, keys, sequence_output, logits, want_gate_mean, sigmoid=False, node_cls=None):
        """egtr:322-418 via the separable algebra.  Returns pre-sigmoid (rel [B,N,N,R], conn [B,N,N,1], gate_mean).
        ``node_cls``: argmax(logits, -1) when the box-decode launch already produced it."""
        bsz, N, d = sequence_output.shape
        unscaling = self.head_dim ** 0.5
        rp, cl = self.rel_predictor.layers, self.connectivity_layer.layers
        wg = self.rel_predictor_gate.weight  # [1, 2d]
        if (ops.inference_fast_path(sequence_output) and ops.GEMM_SPLIT_BF16 and REL_PREP_MERGED and d % 32 == 0
                and (rp[0].weight.shape[0] + cl[0].weight.shape[0]) % 128 == 0 and 2 * (len(queries) + 1) <= 16):
            # Inference: slot projection and first MLP layer are two linear maps in a row -- W1 (P_t x + b_t) =
            # (W1 P_t) x + W1 b_t -- so their product is formed once per weight set (float64, rounded to fp32) and the
            # 2 x (Ld + 1) slot inputs go through ONE grouped split-bf16 launch straight into uq / uk; the gate logits
            # (one output column per slot) follow as one skinny launch.  Three launches -> two, and the [B N T, d]
            # intermediate never exists.  (sqrt(D) un-scaling of egtr:343 folded into the query slots' weights.)
            T = len(queries) + 1
            hd2 = rp[0].weight.shape[0] + cl[0].weight.shape[0]
            projs_q = list(self.proj_q) + [self.final_sub_proj]
            projs_k = list(self.proj_k) + [self.final_obj_proj]
            srcs = [rp[0].weight, cl[0].weight, wg, rp[0].bias, cl[0].bias, self.rel_predictor_gate.bias] + \
                [t for pj in projs_q + projs_k for t in (pj.weight, pj.bias)]

            def build():
                f64 = torch.float64
                w1 = torch.cat([rp[0].weight, cl[0].weight], 0).to(f64)          # [2 Hd, 2 d]
                w1q, w1k, gq, gk = w1[:, :d], w1[:, d:], wg[:, :d].to(f64), wg[:, d:].to(f64)
                wts, bs, gws, gbs = [], [], [], []
                for side, (w1s, gs, pjs) in enumerate(((w1q, gq, projs_q), (w1k, gk, projs_k))):
                    for t, pj in enumerate(pjs):
                        sc = unscaling if (side == 0 and t < T - 1) else 1.0
                        pw, pb = pj.weight.to(f64), pj.bias.to(f64)
                        wts.append(ops.gemm_split_weights(((w1s @ pw) * sc).float().contiguous()))
                        bs.append((w1s @ pb).float().contiguous())
                        gws.append(((gs @ pw) * sc).float().contiguous())        # [1, d]
                        gb = gs @ pb
                        if side == 1:
                            gb = gb + self.rel_predictor_gate.bias.to(f64)
                        gbs.append(gb.float().contiguous())
                return wts, bs, gws, gbs, torch.cat([rp[0].bias, cl[0].bias], 0).contiguous()

            wts, bs, gws, gbs, b1 = ops.cached_weights(self, "rel_head_merged_prep", srcs, build)
            uq = torch.empty(bsz * N, T, hd2, dtype=sequence_output.dtype, device=sequence_output.device)
            uk = torch.empty_like(uq)
            gates = [torch.empty(bsz * N, T, dtype=sequence_output.dtype, device=sequence_output.device) for _ in range(2)]
            xs = [q.transpose(1, 2).reshape(bsz * N, d) for q in queries] + [sequence_output.reshape(bsz * N, d)] + \
                [k.transpose(1, 2).reshape(bsz * N, d) for k in keys] + [sequence_output.reshape(bsz * N, d)]
            outs = [uq[:, t, :] for t in range(T)] + [uk[:, t, :] for t in range(T)]
            ops.linear_split_bf16_grouped([dict(x=x, wt=w, N=hd2, b=b, out=o) for x, w, b, o in zip(xs, wts, bs, outs)])
            gouts = [gates[0][:, t:t + 1] for t in range(T)] + [gates[1][:, t:t + 1] for t in range(T)]
            ops.linear_grouped([dict(x=x, w=w, b=b, out=o) for x, w, b, o in zip(xs, gws, gbs, gouts)])
            uq, uk = uq.view(bsz, N, T, hd2), uk.view(bsz, N, T, hd2)
            gate_q, gate_k = gates[0].view(bsz, N, T), gates[1].view(bsz, N, T)
        elif ops.inference_fast_path(sequence_output):
            # Inference: the 2 x (Ld + 1) slot projections run as ONE grouped launch writing straight into the stacked
            # [B,N,T,d] buffers (the x sqrt(D) unscaling of egtr:343 is the group's input scale), and the four
            # separable first-layer / gate products as one more; their weight slices are cached derived constants.
            T = len(queries) + 1
            Q = torch.empty(bsz, N, T, d, dtype=sequence_output.dtype, device=sequence_output.device)
            K = torch.empty_like(Q)
            Qv, Kv = Q.view(bsz * N, T, d), K.view(bsz * N, T, d)
            items = []
            for t, (q, proj) in enumerate(zip(queries, self.proj_q)):
                items.append(dict(x=q.transpose(1, 2).reshape(bsz, N, d), w=proj.weight, b=proj.bias,
                                  alpha_x=unscaling, out=Qv[:, t, :]))
            items.append(dict(x=sequence_output, w=self.final_sub_proj.weight, b=self.final_sub_proj.bias,
                              out=Qv[:, T - 1, :]))
            for t, (k, proj) in enumerate(zip(keys, self.proj_k)):
                items.append(dict(x=k.transpose(1, 2).reshape(bsz, N, d), w=proj.weight, b=proj.bias,
                                  out=Kv[:, t, :]))
            items.append(dict(x=sequence_output, w=self.final_obj_proj.weight, b=self.final_obj_proj.bias,
                              out=Kv[:, T - 1, :]))
            for i0 in range(0, len(items), 16):
                ops.linear_grouped(items[i0:i0 + 16])
            w1q, w1k, wgq, wgk, b1 = ops.cached_weights(
                self, "rel_head_first_layer",
                [rp[0].weight, cl[0].weight, wg, rp[0].bias, cl[0].bias],
                lambda: (torch.cat([rp[0].weight[:, :d], cl[0].weight[:, :d]], 0).contiguous(),
                         torch.cat([rp[0].weight[:, d:], cl[0].weight[:, d:]], 0).contiguous(),
                         wg[:, :d].contiguous(), wg[:, d:].contiguous(),
                         torch.cat([rp[0].bias, cl[0].bias], 0).contiguous()))
            if ops.GEMM_SPLIT_BF16 and Q.dtype == torch.float32 and d % 32 == 0 and w1q.shape[0] % 128 == 0:
                # [B N T, d] x [d, 2 Hd]: ~1400 rows -- below the token-sized linears' policy threshold, but the split-bf16
                # tile kernel already beats the exact-f32 skinny kernel here (measured 30 -> 12 us with the gate launch)
                wtq, wtk = ops.cached_weights(self, "rel_head_first_layer_split", [rp[0].weight, cl[0].weight],
                                              lambda: (ops.gemm_split_weights(w1q), ops.gemm_split_weights(w1k)))
                uq, uk = ops.linear_split_bf16_grouped([dict(x=Q, wt=wtq, N=w1q.shape[0]),
                                                        dict(x=K, wt=wtk, N=w1k.shape[0])])
                uq, uk = uq.view(bsz, N, T, -1), uk.view(bsz, N, T, -1)
                gate_q, gate_k = ops.linear_grouped([dict(x=Q, w=wgq),
                                                     dict(x=K, w=wgk, b=self.rel_predictor_gate.bias)])
            else:
                uq, uk, gate_q, gate_k = ops.linear_grouped([
                    dict(x=Q, w=w1q), dict(x=K, w=w1k), dict(x=Q, w=wgq),
                    dict(x=K, w=wgk, b=self.rel_predictor_gate.bias)])
            gate_q, gate_k = gate_q[..., 0], gate_k[..., 0]
        else:
            # slot projections q^[b,i,t,:], k^[b,j,t,:]  (t < Ld: decoder layers, t = Ld: final hidden state)
            pq = [ops.module_linear(proj, q.transpose(1, 2).reshape(bsz, N, d) * unscaling)
                  for q, proj in zip(queries, self.proj_q)]
            pk = [ops.module_linear(proj, k.transpose(1, 2).reshape(bsz, N, d)) for k, proj in zip(keys, self.proj_k)]
            pq.append(ops.module_linear(self.final_sub_proj, sequence_output))
            pk.append(ops.module_linear(self.final_obj_proj, sequence_output))
            Q = torch.stack(pq, -2)  # [B,N,T,d]
            K = torch.stack(pk, -2)
            # separable gate logit and first MLP layer:  [W1_rel ; W1_conn ; w_gate] applied to each half
            w1 = torch.cat([rp[0].weight, cl[0].weight], 0)  # [2Hd,2d]
            wq = torch.cat([w1[:, :d], wg[:, :d]], 0)  # [2Hd+1, d]
            wk = torch.cat([w1[:, d:], wg[:, d:]], 0)
            tq = ops.linear(Q, wq)  # [B,N,T,2Hd+1]
            tk = ops.linear(K, wk)
            hd2 = w1.shape[0]
            uq, gate_q = tq[..., :hd2].contiguous(), tq[..., hd2].contiguous()
            uk, gate_k = tk[..., :hd2].contiguous(), (tk[..., hd2] + self.rel_predictor_gate.bias).contiguous()
            b1 = torch.cat([rp[0].bias, cl[0].bias], 0)
        triplet, node = None, None
        if self.config.use_freq_bias:  # egtr:405-413
            triplet = self.triplet_dist
            node = node_cls if node_cls is not None else torch.argmax(logits, dim=-1)
        return ops.relation_head(gate_q, gate_k, uq, uk, b1, rp[1].weight, rp[1].bias, rp[2].weight, rp[2].bias,
                                 cl[1].weight, cl[1].bias, cl[2].weight, cl[2].bias, triplet, node,
                                 want_gate_mean, owner=self, sigmoid=sigmoid)

    def _matcher(self):
        return DeformableDetrHungarianMatcher(
            class_cost=self.config.ce_loss_coefficient, bbox_cost=self.config.bbox_cost,
            giou_cost=self.config.giou_cost, smoothing=self.config.smoothing)

    def _heads(self, outputs, want_gate_mean, labels=None, sigmoid=False, last_level_only=None):
        """Detection heads + relation head on the base model's outputs (egtr:283-418).  Returns
        (logits, pred_boxes, outputs_class, outputs_coord, pred_rel, pred_connectivity, gate_mean, pending_match); the
        relation / connectivity logits are PRE-sigmoid unless ``sigmoid`` (inference: applied in the relation-head
        kernel's epilogue).  With ``labels`` on the GPU the Hungarian cost matrix and its
        copy to the host are enqueued BEFORE the relation head is launched (the matcher needs only logits and boxes), so
        that the host-side assignment overlaps the relation-head kernel instead of idling the GPU.
        ``last_level_only``: run the detection heads on the last decoder level only -- legal when nothing reads the other
        levels (no auxiliary losses downstream).  None = decide here: only ``forward`` without labels qualifies;
        ``forward_tensors`` (whose result feeds ``loss_from_tensors``, auxiliary terms included) always asks for every level
        (under ``torch.no_grad()`` -- a validation loss -- the old rule silently dropped every ``*_i`` term)."""
        if last_level_only is None:
            last_level_only = labels is None
        sequence_output = outputs["last_hidden_state"]
        hidden_states = outputs.intermediate_hidden_states
        init_reference = outputs.init_reference_points
        inter_references = outputs.intermediate_reference_points

        if (last_level_only and labels is None and ops.inference_fast_path(hidden_states)
                and not self.config.with_box_refine and hidden_states.shape[1] > 1):
            # inference: nothing reads the intermediate levels' logits / boxes (they feed the auxiliary losses only,
            # egtr:307-316) -- the heads run on the last level's 200 rows instead of all Ld x 200 (the reference computes
            # every level and drops them)
            init_reference = inter_references[:, -2]
            hidden_states, inter_references = hidden_states[:, -1:], inter_references[:, -1:]
        outputs_class, outputs_coord, node_cls = detection_heads(
            self.config, self.class_embed, self.bbox_embed, hidden_states, init_reference, inter_references,
            want_node_cls=True)
        if not self.config.use_freq_bias:
            node_cls = None
        logits = outputs_class[:, -1]
        pred_boxes = outputs_coord[:, -1]
        if self.config.auxiliary_loss:
            outputs_class = outputs_class[:, : self.config.decoder_layers, ...].permute(1, 0, 2, 3)
            outputs_coord = outputs_coord[:, : self.config.decoder_layers, ...].permute(1, 0, 2, 3)

        pending = None
        if labels is not None and logits.is_cuda:
            pending = self._matcher().prepare({"logits": logits, "pred_boxes": pred_boxes}, labels)

        decoder_attention_queries = outputs["decoder_attention_queries"]
        outputs["decoder_attention_queries"] = None
        decoder_attention_keys = outputs["decoder_attention_keys"]
        outputs["decoder_attention_keys"] = None
        pred_rel, pred_connectivity, gate_mean = self._relation_head(
            decoder_attention_queries, decoder_attention_keys, sequence_output, logits,
            want_gate_mean=want_gate_mean, sigmoid=sigmoid, node_cls=node_cls)
        return logits, pred_boxes, outputs_class, outputs_coord, pred_rel, pred_connectivity, gate_mean, pending

    def forward_tensors(self, pixel_values, pixel_mask):
        """The static-shape part of a TRAINING step as a plain tensors -> tensors function (no host synchronisation, no
        Python objects in or out), so that forward and backward can each be replayed from one HIP graph
        (egtr_amd.runtime.DataParallelTrainer(graph=True)).  ``loss_from_tensors`` turns the result into the loss."""
        outputs = self.model(pixel_values, pixel_mask=pixel_mask, output_attentions=False, output_hidden_states=True,
                             output_attention_states=True, return_dict=True)
        logits, pred_boxes, outputs_class, outputs_coord, pred_rel, pred_connectivity, gate_mean, _ = \
            self._heads(outputs, want_gate_mean=True, last_level_only=not self.config.auxiliary_loss)
        res = (logits, pred_boxes, pred_rel, pred_connectivity, gate_mean)
        if self.config.auxiliary_loss:
            res += (outputs_class, outputs_coord)
        if self.config.two_stage:   # the per-token proposal heads' outputs close the tuple
            res += (outputs.enc_outputs_class, outputs.enc_outputs_coord_logits)
        return res

    def loss_from_tensors(self, tensors, labels):
        """Matcher + SGG loss (egtr:420-505) on the tensors ``forward_tensors`` returns."""
        logits, pred_boxes, pred_rel, pred_connectivity, gate_mean = tensors[:5]
        outputs_class = tensors[5] if self.config.auxiliary_loss else None
        outputs_coord = tensors[6] if self.config.auxiliary_loss else None
        enc = tuple(tensors[-2:]) if self.config.two_stage else None
        loss, loss_dict, _ = self._loss(logits, pred_boxes, pred_rel, pred_connectivity, gate_mean, outputs_class,
                                        outputs_coord, labels, enc_outputs=enc)
        return loss, loss_dict

    def _loss(self, logits, pred_boxes, pred_rel, pred_connectivity, gate_mean, outputs_class, outputs_coord, labels,
              pending_match=None, enc_outputs=None):
        auxiliary_outputs = None
        num_object_queries = logits.shape[1]
        matcher = self._matcher()
        criterion = SceneGraphGenerationLoss(
            matcher=matcher, num_object_queries=num_object_queries, num_classes=self.config.num_labels,
            num_rel_labels=self.config.num_rel_labels, eos_coef=self.config.eos_coefficient,
            losses=["labels", "boxes", "relations", "cardinality", "uncertainty"],
            smoothing=self.config.smoothing, rel_sample_negatives=self.config.rel_sample_negatives,
            rel_sample_nonmatching=self.config.rel_sample_nonmatching, model_training=self.training,
            focal_alpha=self.config.focal_alpha,
            rel_sample_negatives_largest=self.config.rel_sample_negatives_largest,
            rel_sample_nonmatching_largest=self.config.rel_sample_nonmatching_largest)
        criterion.to(logits.device)
        outputs_loss = {"logits": logits, "pred_boxes": pred_boxes, "pred_rel": pred_rel,
                        "pred_connectivity": pred_connectivity}  # pre-sigmoid (egtr:450-454)
        if self.config.auxiliary_loss:
            auxiliary_outputs = self._set_aux_loss(outputs_class, outputs_coord)
            outputs_loss["auxiliary_outputs"] = auxiliary_outputs
        if self.config.two_stage:   # egtr:459-464: (class logits, box logits) of every encoder token
            outputs_loss["enc_outputs"] = {"logits": enc_outputs[0], "pred_boxes": enc_outputs[1].sigmoid()}
        loss_dict = criterion(outputs_loss, labels,
                              matched=matcher.finish(pending_match) if pending_match is not None else None)
        weight_dict = {"loss_ce": self.config.ce_loss_coefficient, "loss_bbox": self.config.bbox_loss_coefficient,
                       "loss_giou": self.config.giou_loss_coefficient,
                       "loss_rel": self.config.rel_loss_coefficient,
                       "loss_connectivity": self.config.connectivity_loss_coefficient}
        base_weights = dict(weight_dict)
        if self.config.auxiliary_loss:
            for i in range(self.config.decoder_layers - 1):
                weight_dict.update({f"{k}_{i}": v for k, v in base_weights.items()})
        if self.config.two_stage:   # egtr:484-488
            weight_dict.update({f"{k}_enc": v for k, v in base_weights.items()})
        loss = sum(loss_dict[k] * weight_dict[k] for k in loss_dict.keys() if k in weight_dict)
        for i in range(self.config.decoder_layers + 1):  # rel_gate_{i} logging (egtr:496-505)
            loss_dict[f"rel_gate_{i}"] = gate_mean[i]
        return loss, loss_dict, auxiliary_outputs

    def forward(self, pixel_values, pixel_mask=None, decoder_attention_mask=None, encoder_outputs=None,
                inputs_embeds=None, decoder_inputs_embeds=None, labels=None, output_attentions=None,
                output_hidden_states=None, output_attention_states=None, return_dict=None):
        return_dict = return_dict if return_dict is not None else self.config.use_return_dict
        outputs = self.model(pixel_values, pixel_mask=pixel_mask, decoder_attention_mask=decoder_attention_mask,
                             encoder_outputs=encoder_outputs, inputs_embeds=inputs_embeds,
                             decoder_inputs_embeds=decoder_inputs_embeds, output_attentions=output_attentions,
                             output_hidden_states=output_hidden_states,
                             output_attention_states=True,  # the relation head needs the retained q / k maps
                             return_dict=True)
        # no loss and no logit adjustment: the final sigmoids (egtr:450-454) run in the relation-head epilogue
        fold_sigmoid = labels is None and not self.config.logit_adjustment and not torch.is_grad_enabled()
        logits, pred_boxes, outputs_class, outputs_coord, pred_rel, pred_connectivity, gate_mean, pending = \
            self._heads(outputs, want_gate_mean=labels is not None, labels=labels, sigmoid=fold_sigmoid)

        loss, loss_dict, auxiliary_outputs = None, None, None
        if labels is not None:
            loss, loss_dict, auxiliary_outputs = self._loss(
                logits, pred_boxes, pred_rel, pred_connectivity, gate_mean, outputs_class, outputs_coord, labels, pending,
                enc_outputs=(outputs.enc_outputs_class, outputs.enc_outputs_coord_logits))

        if self.config.logit_adjustment:  # egtr:509-512
            pred_rel = pred_rel - self.config.logit_adj_tau * self.rel_dist.log().to(pred_rel.device)
        if not fold_sigmoid:
            pred_rel = pred_rel.sigmoid()
            pred_connectivity = pred_connectivity.sigmoid()

        if not return_dict:
            output = (logits, pred_boxes) + ((auxiliary_outputs,) if auxiliary_outputs is not None else ()) \
                + outputs.to_tuple()
            return ((loss, loss_dict) + output) if loss is not None else output
        return DetrSceneGraphGenerationOutput(
            loss=loss, loss_dict=loss_dict, logits=logits, pred_boxes=pred_boxes, pred_rel=pred_rel,
            pred_connectivity=pred_connectivity, auxiliary_outputs=auxiliary_outputs,
            last_hidden_state=outputs.last_hidden_state, decoder_hidden_states=outputs.decoder_hidden_states,
            decoder_attentions=outputs.decoder_attentions, cross_attentions=outputs.cross_attentions,
            encoder_last_hidden_state=outputs.encoder_last_hidden_state,
            encoder_hidden_states=outputs.encoder_hidden_states, encoder_attentions=outputs.encoder_attentions)


class SceneGraphGenerationLoss(nn.Module):
    """Hungarian-matched detection losses + relation BCE with relation smoothing and hard-negative sampling +
    connectivity BCE (egtr:544-1034).  Host-side PyTorch on device tensors, like the reference."""

    def __init__(self, matcher, num_object_queries, num_classes, num_rel_labels, eos_coef, losses, smoothing,
                 rel_sample_negatives, rel_sample_nonmatching, model_training, focal_alpha,
                 rel_sample_negatives_largest, rel_sample_nonmatching_largest):
        super().__init__()
        self.num_object_queries = num_object_queries
        self.num_classes = num_classes
        self.num_rel_labels = num_rel_labels
        self.matcher = matcher
        self.eos_coef = eos_coef
        self.losses = losses
        self.rel_loss = torch.nn.BCEWithLogitsLoss(reduction="none")
        self.rel_sample_negatives = rel_sample_negatives
        self.rel_sample_nonmatching = rel_sample_nonmatching
        self.model_training = model_training
        self.focal_alpha = focal_alpha
        self.rel_sample_negatives_largest = rel_sample_negatives_largest
        self.rel_sample_nonmatching_largest = rel_sample_nonmatching_largest
        self.nonmatching_cost = (-torch.log(torch.tensor(1e-8)) * matcher.class_cost + 4 * matcher.bbox_cost
                                 + 2 * matcher.giou_cost - torch.log(torch.tensor((1.0 / smoothing) - 1.0)))
        self.connectivity_loss = torch.nn.BCEWithLogitsLoss(reduction="none")
        self.force_device_relations = False  # tests: take the sync-light relation-loss path on CPU tensors too

    def loss_labels(self, outputs, targets, indices, matching_costs, num_boxes):
        """Focal classification loss (egtr:611-659)."""
        if "logits" not in outputs:
            raise ValueError("No logits were found in the outputs")
        source_logits = outputs["logits"]
        idx = self._get_src_permutation_idx(indices)
        target_classes_o = torch.cat([t["class_labels"][J] for t, (_, J) in zip(targets, indices)])
        target_classes = torch.full(source_logits.shape[:2], self.num_classes, dtype=torch.int64,
                                    device=source_logits.device)
        target_classes[idx] = target_classes_o.to(source_logits.device)
        onehot = torch.zeros([source_logits.shape[0], source_logits.shape[1], source_logits.shape[2] + 1],
                             dtype=source_logits.dtype, layout=source_logits.layout, device=source_logits.device)
        onehot.scatter_(2, target_classes.unsqueeze(-1), 1)
        loss_ce = sigmoid_focal_loss(source_logits, onehot[:, :, :-1], num_boxes, alpha=self.focal_alpha,
                                     gamma=2) * source_logits.shape[1]
        return {"loss_ce": loss_ce}

    @torch.no_grad()
    def loss_cardinality(self, outputs, targets, indices, matching_costs, num_boxes):
        logits = outputs["logits"]
        tgt_lengths = torch.as_tensor([len(v["class_labels"]) for v in targets], device=logits.device)
        card_pred = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1)
        return {"cardinality_error": F.l1_loss(card_pred.float(), tgt_lengths.float())}

    @torch.no_grad()
    def loss_uncertainty(self, outputs, targets, indices, matching_costs, num_boxes):
        # egtr:671-690: mean of u[s] * u[o] over the non-zero target triplets of the matched block.  Evaluated as
        # (sum over pairs of count * u u^T) / (number of triplets): no nonzero() index list, no host synchronisation.
        num, den = [], []
        for target, index, matching_cost in zip(targets, indices, matching_costs):
            tidx = index[1].to(target["rel"].device)
            cnt = (target["rel"][tidx][:, tidx] != 0).sum(-1).to(matching_cost.dtype)   # [T, T]
            u = matching_cost.sigmoid()
            num.append((cnt * torch.outer(u, u)).sum())
            den.append(cnt.sum())
        return {"uncertainty": torch.stack(num).sum() / torch.stack(den).sum()}

    def loss_boxes(self, outputs, targets, indices, matching_costs, num_boxes):
        assert "pred_boxes" in outputs, "No predicted boxes found in outputs"
        idx = self._get_src_permutation_idx(indices)
        src_boxes = outputs["pred_boxes"][idx]
        target_boxes = torch.cat([t["boxes"][i] for t, (_, i) in zip(targets, indices)], dim=0)
        losses = {"loss_bbox": F.l1_loss(src_boxes, target_boxes, reduction="none").sum() / num_boxes}
        loss_giou = 1 - torch.diag(generalized_box_iou(center_to_corners_format(src_boxes),
                                                       center_to_corners_format(target_boxes)))
        losses["loss_giou"] = loss_giou.sum() / num_boxes
        return losses

    def loss_masks(self, outputs, targets, indices, matching_costs, num_boxes):
        assert "pred_masks" in outputs, "No predicted masks found in outputs"
        src_idx = self._get_src_permutation_idx(indices)
        tgt_idx = self._get_tgt_permutation_idx(indices)
        src_masks = outputs["pred_masks"][src_idx]
        target_masks, _ = nested_tensor_from_tensor_list([t["masks"] for t in targets]).decompose()
        target_masks = target_masks.to(src_masks)[tgt_idx]
        src_masks = F.interpolate(src_masks[:, None], size=target_masks.shape[-2:], mode="bilinear",
                                  align_corners=False)[:, 0].flatten(1)
        target_masks = target_masks.flatten(1).view(src_masks.shape)
        return {"loss_mask": sigmoid_focal_loss(src_masks, target_masks, num_boxes),
                "loss_dice": dice_loss(src_masks, target_masks, num_boxes)}

    def loss_relations(self, outputs, targets, indices, matching_costs, num_boxes):
        """egtr:754-815.  Index tensors are moved to the logits' device once per image (the reference indexes
        device tensors with CPU index tensors, egtr:761-785; same values)."""
        if (self.model_training and outputs["pred_rel"].is_cuda and outputs["pred_rel"].dtype == torch.float32
                and self.rel_sample_negatives is not None and self.rel_sample_nonmatching is not None
                and self.rel_sample_negatives_largest and self.rel_sample_nonmatching_largest
                and not self.force_device_relations
                and all(t["rel"].is_cuda and t["rel"].dtype == torch.float32 for t in targets)):
            # the training configuration (train_egtr.py:514-527): value and gradient in one HIP pass, no host sync
            loss_rel, loss_conn = ops.relation_losses(
                outputs["pred_rel"], outputs["pred_connectivity"], targets, indices, matching_costs,
                float(self.nonmatching_cost), self.rel_sample_negatives, self.rel_sample_nonmatching)
            return {"loss_rel": loss_rel, "loss_connectivity": loss_conn}
        if (self.model_training and (outputs["pred_rel"].is_cuda or self.force_device_relations)
                and (self.rel_sample_negatives is not None or self.rel_sample_nonmatching is not None)
                and (self.rel_sample_negatives is None or self.rel_sample_negatives_largest)
                and (self.rel_sample_nonmatching is None or self.rel_sample_nonmatching_largest)):
            return self._loss_relations_device(outputs, targets, indices, matching_costs)
        losses, connect_losses = [], []
        dev = outputs["pred_rel"].device
        for i, ((src_index, target_index), target, matching_cost) in enumerate(zip(indices, targets, matching_costs)):
            # (the matcher returns device index tensors for device outputs, CPU tensors otherwise)
            full_index = torch.arange(self.num_object_queries, device=src_index.device)
            uniques, counts = torch.cat([full_index, src_index]).unique(return_counts=True)
            full_src_index = torch.cat([src_index, uniques[counts == 1]]).to(dev)
            full_target_index = torch.cat([target_index, torch.arange(len(target_index), self.num_object_queries,
                                                                      device=target_index.device)])
            full_matching_cost = torch.cat([matching_cost, torch.full(
                (self.num_object_queries - len(matching_cost),), float(self.nonmatching_cost),
                device=matching_cost.device)])
            pred_rel = outputs["pred_rel"][i, full_src_index][:, full_src_index]
            fti = full_target_index.to(target["rel"].device)
            target_rel = target["rel"][fti][:, fti]
            rel_index = torch.nonzero(target_rel)
            target_connect = torch.zeros(target_rel.shape[0], target_rel.shape[1], 1, device=target_rel.device)
            target_connect[rel_index[:, 0], rel_index[:, 1]] = 1
            pred_connectivity = outputs["pred_connectivity"][i, full_src_index][:, full_src_index]
            connect_losses.append(self.connectivity_loss(pred_connectivity, target_connect))
            if self.model_training:
                loss = self._loss_relations(pred_rel, target_rel, full_matching_cost, self.rel_sample_negatives,
                                            self.rel_sample_nonmatching)
            else:
                loss = self._loss_relations(pred_rel, target_rel, full_matching_cost, None, None)
            losses.append(loss)
        return {"loss_rel": torch.cat(losses).mean(), "loss_connectivity": torch.stack(connect_losses).mean()}

    def _loss_relations_device(self, outputs, targets, indices, matching_costs):
        """Training-mode relation / connectivity losses (egtr:754-923 with ``*_largest`` sampling) without the
        reference's per-image ``nonzero()`` index lists (~2 M x 3 int64 per image for the non-matching candidates) and
        their host synchronisations -- ONE batched synchronisation per step for the candidate counts.  Same value and
        gradient: both losses are means over a set of elements, so they are evaluated in the un-permuted query order
        as (sum of selected BCE terms) / (number of selected terms); the hard negatives are the top-k scores of the
        candidate sets, selected with a masked ``topk`` instead of ``topk`` over gathered index lists (ties have equal
        logits and zero targets, hence equal loss terms)."""
        pred_rel_all, pred_conn_all = outputs["pred_rel"], outputs["pred_connectivity"]
        dev = pred_rel_all.device
        N, R = self.num_object_queries, self.num_rel_labels
        nmc = float(self.nonmatching_cost)
        per_image = []
        counts = []
        for (src_index, target_index), target, matching_cost in zip(indices, targets, matching_costs):
            T = int(len(src_index))
            sidx, tidx = src_index.to(dev), target_index.to(dev)
            # target row of every query: matched -> its target, unmatched (ascending) -> rows T, T+1, ... (egtr:761-768)
            unmatched = torch.ones(N, dtype=torch.bool, device=dev)
            unmatched[sidx] = False
            tq = T + torch.cumsum(unmatched, 0) - 1          # unmatched queries, ascending -> rows T, T+1, ...
            tq[sidx] = tidx
            rel_t = target["rel"]
            block_t = rel_t[tidx][:, tidx]                       # [T, T, R] targets of the matched block
            counts.append(torch.stack([(block_t != 0).sum(), (block_t != 1.0).sum()]))
            per_image.append((T, sidx, tidx, tq, block_t, matching_cost))
        cnt = torch.stack(counts).tolist() if counts else []     # the one host synchronisation
        rel_sum, rel_cnt, conn_losses = [], 0, []
        for i, (T, sidx, tidx, tq, block_t, matching_cost) in enumerate(per_image):
            n_true, n_false = int(cnt[i][0]), int(cnt[i][1])
            pred = pred_rel_all[i]                               # [N, N, R]
            cost_q = torch.full((N,), nmc, device=dev, dtype=matching_cost.dtype)
            cost_q[sidx] = matching_cost
            wq = 1.0 - cost_q.sigmoid()
            rel_t = targets[i]["rel"]
            # connectivity (egtr:786-793): target = any non-zero relation between the two queries' target rows
            tc = (rel_t != 0).any(-1).to(pred.dtype)             # [N, N] over target rows
            conn_losses.append(self.connectivity_loss(pred_conn_all[i], tc[tq][:, tq].unsqueeze(-1)))
            # matched block
            block_p = pred[sidx][:, sidx]                        # [T, T, R]
            wb = wq[sidx]
            wblock = (wb[:, None] * wb[None, :]).unsqueeze(-1)
            l_block = self.rel_loss(block_p, block_t * wblock)
            parts = [(l_block * (block_t != 0)).sum()]
            n_sel = n_true
            false_mask = block_t != 1.0
            if self.rel_sample_negatives is None:
                parts.append((l_block * false_mask).sum())
                n_sel += n_false
            else:
                k1 = min(n_true * self.rel_sample_negatives, n_false) if n_true > 0 else 0
                if k1 > 0:
                    sel = torch.topk(block_p.detach().masked_fill(~false_mask, float("-inf")).reshape(-1), k1)[1]
                    parts.append(self.rel_loss(block_p.reshape(-1)[sel],
                                               (block_t * wblock).reshape(-1)[sel]).sum())
                    n_sel += k1
            # non-matching region: every (a, b, r) with a or b unmatched
            mq = torch.zeros(N, dtype=torch.bool, device=dev)
            mq[sidx] = True
            nm_mask = ~(mq[:, None] & mq[None, :])               # [N, N]
            n_nm = (N * N - T * T) * R
            if self.rel_sample_nonmatching is None:
                tgt_full = rel_t[tq][:, tq] * (wq[:, None] * wq[None, :]).unsqueeze(-1)
                parts.append((self.rel_loss(pred, tgt_full) * nm_mask.unsqueeze(-1)).sum())
                n_sel += n_nm
            else:
                k2 = min(n_true * self.rel_sample_nonmatching, n_nm) if n_true > 0 else 0
                if k2 > 0:
                    sel = torch.topk(pred.detach().masked_fill(~nm_mask.unsqueeze(-1), float("-inf")).reshape(-1),
                                     k2)[1]
                    a = torch.div(sel, N * R, rounding_mode="floor")
                    b = torch.div(sel, R, rounding_mode="floor") % N
                    r = sel % R
                    tgt = rel_t[tq[a], tq[b], r] * (wq[a] * wq[b])
                    parts.append(self.rel_loss(pred.reshape(-1)[sel], tgt).sum())
                    n_sel += k2
            rel_sum.append(torch.stack(parts).sum())
            rel_cnt += n_sel
        if rel_cnt > 0:
            loss_rel = torch.stack(rel_sum).sum() / rel_cnt
        else:  # no relation in the batch: the reference takes the mean of an empty tensor
            loss_rel = torch.cat([pred_rel_all.new_zeros(0)]).mean()
        return {"loss_rel": loss_rel, "loss_connectivity": torch.stack(conn_losses).mean()}

    def _loss_relations(self, pred_rel, target_rel, matching_cost, rel_sample_negatives, rel_sample_nonmatching):
        """egtr:817-923."""
        if (rel_sample_negatives is None) and (rel_sample_nonmatching is None):
            weight = 1.0 - matching_cost.sigmoid()
            weight = torch.outer(weight, weight)
            target_rel = target_rel * weight.unsqueeze(-1)
            return self.rel_loss(pred_rel, target_rel).mean(-1).reshape(-1)
        matched = matching_cost != self.nonmatching_cost.to(matching_cost.device)
        num_target_objects = int(matched.sum())
        sub = target_rel[:num_target_objects, :num_target_objects, :]
        true_indices = sub.nonzero()
        false_indices = (sub != 1.0).nonzero()
        nonmatching_indices = (torch.outer(matched, matched).unsqueeze(-1).repeat(1, 1, self.num_rel_labels)
                               != True).nonzero()  # noqa: E712
        num_target_relations = len(true_indices)

        def _sample(cands, k, largest_flag):
            if k == 0 or num_target_relations == 0:
                return cands[[]]
            n = min(num_target_relations * k, cands.size(0))
            if largest_flag:
                scores = pred_rel[cands[:, 0], cands[:, 1], cands[:, 2]]
                sel = torch.topk(scores, n, largest=True)[1]
            else:
                sel = torch.tensor(random.sample(range(cands.size(0)), n), device=cands.device)
            return cands[sel]

        if rel_sample_negatives is not None:
            false_indices = _sample(false_indices, rel_sample_negatives, self.rel_sample_negatives_largest)
        if rel_sample_nonmatching is not None:
            nonmatching_indices = _sample(nonmatching_indices, rel_sample_nonmatching,
                                          self.rel_sample_nonmatching_largest)
        relation_indices = torch.cat([true_indices, false_indices, nonmatching_indices])
        pred_rel = pred_rel[relation_indices[:, 0], relation_indices[:, 1], relation_indices[:, 2]]
        target_rel = target_rel[relation_indices[:, 0], relation_indices[:, 1], relation_indices[:, 2]]
        weight = 1.0 - matching_cost.sigmoid()
        weight = weight[relation_indices[:, 0]] * weight[relation_indices[:, 1]]
        return self.rel_loss(pred_rel, target_rel * weight)

    def _get_src_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        src_idx = torch.cat([src for (src, _) in indices])
        return batch_idx, src_idx

    def _get_tgt_permutation_idx(self, indices):
        batch_idx = torch.cat([torch.full_like(tgt, i) for i, (_, tgt) in enumerate(indices)])
        tgt_idx = torch.cat([tgt for (_, tgt) in indices])
        return batch_idx, tgt_idx

    def get_loss(self, loss, outputs, targets, indices, matching_costs, num_boxes):
        loss_map = {"labels": self.loss_labels, "cardinality": self.loss_cardinality, "boxes": self.loss_boxes,
                    "masks": self.loss_masks, "relations": self.loss_relations,
                    "uncertainty": self.loss_uncertainty}
        assert loss in loss_map, f"Loss {loss} not supported"
        return loss_map[loss](outputs, targets, indices, matching_costs, num_boxes)

    def forward(self, outputs, targets, matched=None):
        """egtr:953-1034.  ``num_boxes`` is per-rank (the reference's all-reduce is commented out, :976-980).
        ``matched``: the (indices, matching_costs) of the main outputs when the caller already ran the matcher."""
        outputs_without_aux = {k: v for k, v in outputs.items() if k not in ("auxiliary_outputs", "enc_outputs")}
        indices, matching_costs = matched if matched is not None else self.matcher(outputs_without_aux, targets)
        num_boxes = float(max(sum(len(t["class_labels"]) for t in targets), 1))
        losses = {}
        # labels + cardinality + boxes of an output set as ONE HIP launch (values and gradients) when the device matcher
        # packed the indices: ~40 small tensor ops per set otherwise, 7 sets with the auxiliary losses
        fused = ("labels", "cardinality", "boxes")
        packed = None

        def fused_ok(out, idx):
            lg = out.get("logits")
            return (all(k in self.losses for k in fused) and getattr(idx, "flat", None) is not None and lg is not None
                    and lg.is_cuda and lg.dtype == torch.float32 and out["pred_boxes"].dtype == torch.float32
                    and lg.shape[1] <= 2048)

        def detection(out, idx):
            nonlocal packed
            if packed is None:
                packed = ops.pack_detection_targets(targets, out["logits"].device)
            return ops.detection_losses(out["logits"], out["pred_boxes"], idx.flat, packed, self.focal_alpha, num_boxes)

        def poisoned(idx, terms):
            # a cost matrix the device matcher refused (NaN / -inf entries; scipy raises there): every loss term of THAT
            # output set comes out NaN -- the kernels skip a refused image, so without this its terms would look healthy
            # in a logged loss_dict (the ValueError itself follows at the next host synchronisation point)
            return idx.poison_terms(terms) if isinstance(idx, MatchedIndices) else terms

        use_fused = fused_ok(outputs, indices)
        main = {}
        if use_fused:
            main.update(detection(outputs, indices))
        for loss in self.losses:
            if use_fused and loss in fused:
                continue
            main.update(self.get_loss(loss, outputs, targets, indices, matching_costs, num_boxes))
        losses.update(poisoned(indices, main))
        if "auxiliary_outputs" in outputs:
            for i, auxiliary_outputs in enumerate(outputs["auxiliary_outputs"]):
                indices, matching_costs = self.matcher(auxiliary_outputs, targets)
                aux_fused = fused_ok(auxiliary_outputs, indices)
                aux = {}
                if aux_fused:
                    aux.update({k + f"_{i}": v for k, v in detection(auxiliary_outputs, indices).items()})
                for loss in self.losses:
                    if loss in ["masks", "relations", "uncertainty"] or (aux_fused and loss in fused):
                        continue
                    l_dict = self.get_loss(loss, auxiliary_outputs, targets, indices, matching_costs, num_boxes)
                    aux.update({k + f"_{i}": v for k, v in l_dict.items()})
                losses.update(poisoned(indices, aux))
        if "enc_outputs" in outputs:   # two-stage proposals against class-agnostic targets (egtr:1019-1033)
            enc_outputs = outputs["enc_outputs"]
            bin_targets = [dict(t, class_labels=torch.zeros_like(t["class_labels"])) for t in targets]
            indices, matching_costs = self.matcher(enc_outputs, bin_targets)
            enc = {}
            for loss in self.losses:
                if loss in ["masks", "relations", "uncertainty"]:
                    continue
                l_dict = self.get_loss(loss, enc_outputs, bin_targets, indices, matching_costs, num_boxes)
                enc.update({k + "_enc": v for k, v in l_dict.items()})
            losses.update(poisoned(indices, enc))
        return losses
