"""ctypes loader for libtrackie_mi355x.so (in-tree build; fails loudly when absent)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TK_MI355X_LIB", os.path.join(HERE, "libtrackie_mi355x.so"))  # override: diagnostic builds only
_lib = None


class TkError(RuntimeError):
    def __init__(self, code, detail):
        super().__init__(f"tk error {code}: {detail}")
        self.code = code
        self.detail = detail


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C trackiellm_amd/csrc -j8` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.tk_error_get_detail.restype = C.c_char_p
        _lib.tk_error_to_string.restype = C.c_char_p
        _lib.tk_mi355x_version.restype = C.c_char_p
    return _lib


def check(rc):
    if rc != 0:
        raise TkError(rc, lib().tk_error_get_detail().decode("utf-8", "replace"))
