"""Elasticity scaling of candidate widths to a latency target (epoch-boundary host logic;
contract: train_search.py:478-532 ``fit_mc_num_by_latency`` / ``bound_clip``).

SURVEY.md section 8(f) lists this as the first "next" row after the inner loop; the host arithmetic is small
and pinned by tests/golden/latency_kat.json (captured from the reference's own functions).
"""
import copy

from .latency import get_lookup_latency


def bound_clip(mc_num, max_mc_num):
    """Clamp a width into [max//2, max]; the flag says whether it can still move."""
    lo = max_mc_num // 2
    if mc_num <= lo:
        return lo, False
    if mc_num >= max_mc_num:
        return max_mc_num, False
    return mc_num, True


def fit_mc_num_by_latency(parsed_arch, mc_num_dddict, mc_maxnum_dddict, lat_lookup_key_dddict, lat_lookup,
                          target_lat, stages, sign):
    """Grow (sign=+1) or shrink (sign=-1) the widths of the chosen ops, in proportion to their current
    ratios, until the looked-up latency crosses ``target_lat`` or every width hits its bound."""
    assert sign in (-1, 1)
    chosen = [(st, blk, parsed_arch[st][blk]) for st in stages for blk in parsed_arch[st]]
    cur = [mc_num_dddict[st][blk][op] for st, blk, op in chosen]
    caps = [mc_maxnum_dddict[st][blk][op] for st, blk, op in chosen]
    unit = min(cur)
    steps = [int(round(c / unit)) for c in cur]
    movable = [True] * len(chosen)

    lat = get_lookup_latency(parsed_arch, mc_num_dddict, lat_lookup_key_dddict, lat_lookup)
    trial, trial_lat = copy.deepcopy(mc_num_dddict), lat
    while any(movable) and sign * trial_lat <= sign * target_lat:
        mc_num_dddict, lat = copy.deepcopy(trial), trial_lat
        for j, (st, blk, op) in enumerate(chosen):
            trial[st][blk][op], movable[j] = bound_clip(mc_num_dddict[st][blk][op] + sign * steps[j], caps[j])
        trial_lat = get_lookup_latency(parsed_arch, trial, lat_lookup_key_dddict, lat_lookup)
    if sign == -1:       # shrinking keeps the first configuration at/below the target
        mc_num_dddict, lat = copy.deepcopy(trial), trial_lat
    return mc_num_dddict, lat
