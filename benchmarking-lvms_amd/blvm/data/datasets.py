"""Dataset registry with the reference's shape (blvm/data/datasets.py:7-55): name -> namespace of source-file paths under the
data root (`BLVM_DATA_ROOT_DIRECTORY`, default ./data), audio extension and length column.  The source CSVs are produced by the
reference's `scripts/data/prepare_*.py`; decoding FLAC needs torchaudio (WAV copies work without)."""
import os
from types import SimpleNamespace

DATA_DIRECTORY = os.environ.get("BLVM_DATA_ROOT_DIRECTORY", os.path.join(os.getcwd(), "data"))


def _src(*parts):
    return os.path.join(DATA_DIRECTORY, "source_files", *parts)


TIMIT, LIBRISPEECH = "timit", "librispeech"

DATASETS = {
    TIMIT: SimpleNamespace(name=TIMIT, train=_src(TIMIT, "train"), valid=_src(TIMIT, "valid"), test=_src(TIMIT, "test"),
                           valid_sets=[_src(TIMIT, "valid")], test_sets=[_src(TIMIT, "test")], audio_ext="flac",
                           audio_length="length.flac.samples"),
    LIBRISPEECH: SimpleNamespace(name=LIBRISPEECH, train=_src(LIBRISPEECH, "train"), valid=_src(LIBRISPEECH, "dev-clean"),
                                 test=_src(LIBRISPEECH, "test-clean"), valid2=_src(LIBRISPEECH, "dev-other"),
                                 test2=_src(LIBRISPEECH, "test-other"),
                                 valid_sets=[_src(LIBRISPEECH, "dev-clean"), _src(LIBRISPEECH, "dev-other")],
                                 test_sets=[_src(LIBRISPEECH, "test-clean"), _src(LIBRISPEECH, "test-other")], audio_ext="flac",
                                 audio_length="length.flac.samples"),
}  # fmt: skip
