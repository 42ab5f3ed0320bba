"""Generate tests/golden/reference_state_spec.npz: the NAMES, SHAPES and DTYPES of the reference model's state_dict as a
training checkpoint holds it (train.py:181-190: `model.state_dict()` of the SyncBatchNorm-converted, DDP-wrapped
PQ_Transformer, hence the `module.` prefix), and the two AdamW parameter groups of train.py:363-374 (names containing
"decoder" get their own learning rate).  DATA only -- no weights.  The reference model is instantiated on the CPU exactly as
tests/golden/make_golden.py does (its import preamble is reused).

    python tests/golden/make_golden_state_spec.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import make_golden  # noqa: E402  (sets up the import of the reference's model files; nothing runs)


def main():
    import pq_transformer as ref_model
    assert ref_model.__file__.startswith(make_golden.REF)
    out = {}
    for extra in (0, 6):
        net = ref_model.PQ_Transformer(input_feature_dim=extra, num_class=18, num_proposal=256, num_quad_proposal=256,
                                       num_heading_bin=1, num_size_cluster=18, mean_size_arr=make_golden.mean_size_arr())
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)                 # pq_transformer.py:194 / train.py
        sd = net.state_dict()
        tag = f"c{extra}"
        out[f"{tag}.names"] = np.array(["module." + k for k in sd.keys()])
        out[f"{tag}.shapes"] = np.array([",".join(str(d) for d in v.shape) for v in sd.values()])
        out[f"{tag}.dtypes"] = np.array([str(v.dtype) for v in sd.values()])
        # modules registered under two names (the position embeddings: pq_transformer.py:186-189) appear twice with the
        # same storage: alias_of[i] = index of the first entry that shares entry i's storage
        first = {}
        out[f"{tag}.alias_of"] = np.array([first.setdefault((v.data_ptr(), tuple(v.shape)), i) if v.numel() else i
                                           for i, v in enumerate(sd.values())], dtype=np.int64)
        named = [n for n, p in net.named_parameters() if p.requires_grad]
        out[f"{tag}.group_plain"] = np.array([n for n in named if "decoder" not in n])
        out[f"{tag}.group_decoder"] = np.array([n for n in named if "decoder" in n])
        print(f"{tag}: {len(sd)} entries ({int((out[f'{tag}.alias_of'] != np.arange(len(sd))).sum())} aliases), "
              f"{sum(v.numel() for v in sd.values())} values, "
              f"{len(out[f'{tag}.group_plain'])} + {len(out[f'{tag}.group_decoder'])} parameters")
    path = os.path.join(HERE, "reference_state_spec.npz")
    np.savez_compressed(path, **out)
    print(f"reference_state_spec.npz: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
