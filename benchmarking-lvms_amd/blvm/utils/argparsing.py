"""`str2bool` under the module name the reference's entry points import it from (blvm/utils/argparsing.py)."""
from blvm.utils.argparsers import str2bool  # noqa: F401
