"""The reference's Mha layer as ONE node (layers/transformer/mha.rs:633-724: final_mul, softmax, qk under one node id, one MhaProof) on the device:
golden cases 11 / 12 of tests/golden/graph_models.json (models.mha_block: LayerNorm -> QKV -> Mha -> projection -> residual) and 13
(models.transformer_layer: a whole pre-LN layer, attention and feed-forward half, as one graph of 19 nodes). Same checks as the
other graph cases (tests/test_gpu_model.py): bytes equal to the oracle's and to the committed sha256, verifier verdicts, batch = sequential.
(Its own file, sorted last: the node kind has not run on hardware before this round's end.)"""
import pytest

from test_gpu_model import test_graph_model_proof_bytes_identical_to_oracle_and_golden as _graph_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [11, 12, 13])
def test_mha_block_proof_bytes_identical_to_oracle_and_golden(dev, oracle, case):
    _graph_case(dev, oracle, case)
