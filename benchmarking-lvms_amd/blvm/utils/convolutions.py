"""Receptive-field / stride arithmetic of stacked convolutions (host-side shape logic), with the reference's names
(blvm/utils/convolutions.py:83-125)."""


def compute_conv_attributes_single(i=0, k=float("nan"), p=float("nan"), s=float("nan"), d=1, s_in=1, r_in=1, start_in=0):
    """Map (feature count, feature spacing, receptive field, first-feature centre) through one conv layer of kernel `k`,
    padding `p`, stride `s`, dilation `d`.  Returns (o_out, s_out, r_out, start_out)."""
    k_eff = d * (k - 1) + 1
    o_out = (i + 2 * p - k_eff) // s + 1
    s_out = s_in * s
    r_out = r_in + (k_eff - 1) * s_in
    total_padding = (o_out - 1) * s + k_eff - i
    start_out = start_in + ((k_eff - 1) / 2 - total_padding // 2) * s_in
    return o_out, s_out, r_out, start_out
