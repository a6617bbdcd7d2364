"""The similarity / zero-shot entry points of the reference's CLIP-style API
(/root/reference/gloria/gloria.py:184-275) on the MI355X kernels.  Checkpoint download / loading helpers,
prompt generation and the classification / segmentation model loaders of that file are out of scope
(SURVEY.md section 2 row 15); build the model with `gloria.builder.build_gloria_model(cfg)` or
`build_gloria_from_ckpt`.

Inputs are ALREADY processed tensors, as in the reference (`process_img` / `process_text` are host preprocessing):
imgs float [B, 3, H, W]; txts a dict with caption_ids / attention_mask / token_type_ids / cap_lens."""

import numpy as np
import torch


def get_similarities(gloria_model, imgs, txts, similarity_type="both"):
    """Similarities between every image and every text, numpy [B_img, B_txt]  (ref :184-240):
    global = cosine (K3), local = max-over-words attention-weighted similarity (K1, words 1..n, temps 4 / 5),
    both = their mean."""
    if similarity_type not in ["global", "local", "both"]:
        raise RuntimeError("similarity type should be one of ['global', 'local', 'both']")
    if type(txts) == str or type(txts) == list:
        raise RuntimeError("Text input not processed - please use gloria_model.process_text")
    if type(imgs) == str or type(imgs) == list:
        raise RuntimeError("Image input not processed - please use gloria_model.process_img")
    with torch.no_grad():
        img_emb_l, img_emb_g = gloria_model.image_encoder_forward(imgs)
        text_emb_l, text_emb_g, _ = gloria_model.text_encoder_forward(
            txts["caption_ids"], txts["attention_mask"], txts["token_type_ids"])
    global_similarities = gloria_model.get_global_similarities(img_emb_g, text_emb_g)
    local_similarities = gloria_model.get_local_similarities(img_emb_l, text_emb_l, txts["cap_lens"])
    similarities = (local_similarities + global_similarities) / 2
    if similarity_type == "global":
        return global_similarities.detach().cpu().numpy()
    elif similarity_type == "local":
        return local_similarities.detach().cpu().numpy()
    return similarities.detach().cpu().numpy()


def normalize(similarities, method="norm"):
    """ref gloria/utils/utils.py:12-21"""
    if method == "norm":
        return (similarities - similarities.mean(axis=0)) / (similarities.std(axis=0))
    elif method == "standardize":
        return (similarities - similarities.min(axis=0)) / (similarities.max(axis=0) - similarities.min(axis=0))
    raise Exception("normalizing method not implemented")


def zero_shot_classification(gloria_model, imgs, cls_txt_mapping):
    """Per class the best prompt's similarity, normalised across images (ref :243-275).  Returns
    (class_similarities [B_img, n_classes] numpy, class names) - the reference wraps the same array in a
    pandas DataFrame with these column names."""
    class_similarities = []
    for cls_name, cls_txt in cls_txt_mapping.items():
        similarities = get_similarities(gloria_model, imgs, cls_txt, similarity_type="both")
        class_similarities.append(similarities.max(axis=1))
    class_similarities = np.stack(class_similarities, axis=1)
    if class_similarities.shape[0] > 1:
        class_similarities = normalize(class_similarities)
    return class_similarities, list(cls_txt_mapping.keys())
