"""ctypes binding of oracle/liboracle.so — the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package image_matching_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

def auto_babies(dim, blocks):
    """Mirror of the product's auto rule (include/hydia.h hydia_auto_babies; asserted equal in tests/test_gpu_parity.py): how many
    hoisted (baby) rotations of the query a database of `blocks` 16384-vector blocks is enrolled for.  dim = the reference's own form."""
    base = 1
    while base * base < dim:
        base *= 2
    for limit, mult in ((3, 2), (12, 4), (40, 8)):
        if blocks <= limit:
            return min(dim, base * mult)
    return dim


u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    # GPU boxes expose every hardware thread but grant a 16-core share: unbounded OpenMP teams oversubscribe badly
    os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp = C.c_void_p
    sig = {
        "hyo_params_create": (vp, [C.c_int] * 6),
        "hyo_params_create_custom": (vp, [C.c_int] * 6 + [vp, vp]),
        "hyo_params_free": (None, [vp]),
        "hyo_get_moduli": (None, [vp, vp]),
        "hyo_get_roots": (None, [vp, vp]),
        "hyo_get_info": (C.c_int, [vp, vp]),
        "hyo_ntt_fwd": (None, [vp, vp, C.c_int]),
        "hyo_ntt_inv": (None, [vp, vp, C.c_int]),
        "hyo_galois_elt": (C.c_uint64, [vp, C.c_int]),
        "hyo_automorph_eval": (None, [vp, vp, vp, C.c_uint64]),
        "hyo_automorph_coeff": (None, [vp, vp, vp, C.c_uint64, C.c_uint64]),
        "hyo_chacha_block": (None, [vp, C.c_uint64, C.c_uint64, vp]),
        "hyo_sample_uniform": (None, [vp, C.c_uint64, C.c_uint64, vp, C.c_int]),
        "hyo_sample_ternary": (None, [vp, C.c_uint64, vp, C.c_int]),
        "hyo_sample_gauss": (None, [vp, C.c_uint64, vp, C.c_int]),
        "hyo_encode": (None, [vp, vp, C.c_int, C.c_double, C.c_int, vp]),
        "hyo_encode_coeffs": (None, [vp, vp, C.c_int, C.c_double, vp]),
        "hyo_decode": (None, [vp, vp, C.c_int, C.c_double, vp]),
        "hyo_keygen": (vp, [vp, vp, vp, C.c_int]),
        "hyo_keys_free": (None, [vp]),
        "hyo_keys_rot": (vp, [vp, C.c_int]),
        "hyo_ct_alloc": (vp, [vp, C.c_int, C.c_int, C.c_double]),
        "hyo_ct_clone": (vp, [vp, vp]),
        "hyo_ct_free": (None, [vp]),
        "hyo_ct_data": (vp, [vp]),
        "hyo_ct_nl": (C.c_int, [vp]),
        "hyo_ct_npoly": (C.c_int, [vp]),
        "hyo_ct_scale": (C.c_double, [vp]),
        "hyo_encrypt": (vp, [vp, vp, vp, C.c_int, vp, C.c_uint64]),
        "hyo_decrypt": (None, [vp, vp, vp, vp]),
        "hyo_keyswitch": (None, [vp, vp, C.c_int, vp, vp, vp]),
        "hyo_hoist_precompute": (vp, [vp, vp, C.c_int]),
        "hyo_rotate_hoisted": (vp, [vp, vp, vp, vp, C.c_int]),
        "hyo_rotate": (vp, [vp, vp, vp, C.c_int]),
        "hyo_mult_norelin": (vp, [vp, vp, vp]),
        "hyo_add_inplace": (None, [vp, vp, vp]),
        "hyo_sub_inplace": (None, [vp, vp, vp]),
        "hyo_relin_inplace": (None, [vp, vp, vp]),
        "hyo_rescale_inplace": (None, [vp, vp]),
        "hyo_drop_to": (None, [vp, vp, C.c_int]),
        "hyo_add_const": (None, [vp, vp, C.c_double]),
        "hyo_mul_const": (vp, [vp, vp, C.c_double, C.c_double]),
        "hyo_mult": (vp, [vp, vp, vp, vp]),
        "hyo_chebyshev_step_coeffs": (None, [C.c_double, C.c_int, vp]),
        "hyo_compare_plain": (C.c_double, [C.c_double, C.c_double, C.c_int]),
        "hyo_eval_chebyshev63": (vp, [vp, vp, vp, vp, C.c_int]),
        "hyo_eval_f4": (vp, [vp, vp, vp]),
        "hyo_chebyshev_compare": (vp, [vp, vp, vp, C.c_double, C.c_int]),
        "hyo_normalize": (None, [vp, C.c_int]),
        "hyo_enroll_num_cts": (C.c_size_t, [vp, C.c_size_t]),
        "hyo_enroll_layout_row": (None, [vp, vp, C.c_size_t, C.c_size_t, vp]),
        "hyo_enroll": (vp, [vp, vp, vp, C.c_size_t, vp, vp]),
        "hyo_encrypt_query": (vp, [vp, vp, vp, vp, C.c_uint64]),
        "hyo_rotate_query": (vp, [vp, vp, vp]),
        "hyo_similarity_block": (vp, [vp, vp, vp, vp]),
        "hyo_compute_similarity": (vp, [vp, vp, vp, vp, C.c_size_t, vp]),
        "hyo_index_scenario": (vp, [vp, vp, vp, vp, C.c_size_t, vp]),
        "hyo_membership_scenario": (vp, [vp, vp, vp, vp, C.c_size_t]),
        "hyo_bsgs_babies": (C.c_int, [vp]),
        "hyo_enroll_layout_row_bsgs": (None, [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_int]),
        "hyo_enroll_bsgs": (vp, [vp, vp, vp, C.c_size_t, vp, vp, C.c_int]),
        "hyo_compute_similarity_bsgs": (vp, [vp, vp, vp, vp, C.c_size_t, vp, C.c_int]),
        "hyo_index_scenario_bsgs": (vp, [vp, vp, vp, vp, C.c_size_t, vp, C.c_int]),
        "hyo_membership_scenario_bsgs": (vp, [vp, vp, vp, vp, C.c_size_t, C.c_int]),
        "hyo_db_write_files": (C.c_int, [vp, vp, C.c_size_t, C.c_char_p]),
        "hyo_index_scenario_files": (vp, [vp, vp, vp, C.c_char_p, C.c_size_t, vp]),
        "hyo_decrypt_membership": (C.c_int, [vp, vp, vp]),
        "hyo_decrypt_index": (C.c_size_t, [vp, vp, vp, C.c_size_t, vp, C.c_size_t]),
        "hyo_hers_layout_row": (None, [vp, vp, C.c_size_t, C.c_size_t, vp]),
        "hyo_hers_enroll": (vp, [vp, vp, vp, C.c_size_t, vp, vp]),
        "hyo_hers_encrypt_query": (vp, [vp, vp, vp, vp, C.c_uint64]),
        "hyo_hers_compute_similarity": (vp, [vp, vp, vp, vp, C.c_size_t, vp]),
        "hyo_hers_index_scenario": (vp, [vp, vp, vp, vp, C.c_size_t, vp]),
        "hyo_hers_membership_scenario": (vp, [vp, vp, vp, vp, C.c_size_t]),
        "hyo_ct_at": (vp, [vp, C.c_size_t]),
        "hyo_ct_array_free": (None, [vp, C.c_size_t]),
        "hyo_num_threads": (C.c_int, []),
        "hyo_set_num_threads": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def seed_bytes(x):
    """32-byte ChaCha key from an int or bytes."""
    if isinstance(x, (bytes, bytearray)):
        b = bytes(x)
        assert len(b) == 32
    else:
        b = int(x).to_bytes(32, "little")
    return np.frombuffer(b, dtype=np.uint8).copy()


class Params:
    def __init__(self, log_n=15, depth=11, scale_bits=45, first_bits=60, dnum=3, dim=512, moduli=None, roots=None, n_p=None):
        self.L = lib()
        if moduli is None:
            self.h = self.L.hyo_params_create(log_n, depth, scale_bits, first_bits, dnum, dim)
        else:  # caller-supplied prime chain: n_q ciphertext primes (q_0 first) then n_p special primes
            moduli = np.ascontiguousarray(moduli, dtype=np.uint64)
            roots = None if roots is None else np.ascontiguousarray(roots, dtype=np.uint64)
            self.h = self.L.hyo_params_create_custom(log_n, len(moduli) - n_p, n_p, scale_bits, dnum, dim, _ptr(moduli),
                                                     None if roots is None else _ptr(roots))
            if not self.h:
                raise ValueError("oracle: custom moduli rejected")
        info = np.zeros(8, dtype=np.int32)
        self.L.hyo_get_info(self.h, _ptr(info))
        self.log_n, self.N, self.nQ, self.nP, self.dnum, self.alpha, self.dim, self.slots = [int(v) for v in info]
        self.nT = self.nQ + self.nP
        self.moduli = np.zeros(self.nT, dtype=np.uint64)
        self.L.hyo_get_moduli(self.h, _ptr(self.moduli))
        self.roots = np.zeros(self.nT, dtype=np.uint64)
        self.L.hyo_get_roots(self.h, _ptr(self.roots))
        self.delta = float(2.0 ** scale_bits)

    def close(self):
        if self.h:
            self.L.hyo_params_free(self.h)
            self.h = None

    # ---- primitives
    def ntt_fwd(self, a, m):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.L.hyo_ntt_fwd(self.h, _ptr(a), m)
        return a

    def ntt_inv(self, a, m):
        a = np.ascontiguousarray(a, dtype=np.uint64).copy()
        self.L.hyo_ntt_inv(self.h, _ptr(a), m)
        return a

    def galois(self, rot):
        return int(self.L.hyo_galois_elt(self.h, rot))

    def automorph_eval(self, a, g):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        o = np.empty_like(a)
        self.L.hyo_automorph_eval(self.h, _ptr(a), _ptr(o), g)
        return o

    def automorph_coeff(self, a, g, q):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        o = np.empty_like(a)
        self.L.hyo_automorph_coeff(self.h, _ptr(a), _ptr(o), g, int(q))
        return o

    def encode(self, slots, scale=None, nl=None):
        slots = np.ascontiguousarray(slots, dtype=np.float64)
        nl = self.nQ if nl is None else nl
        out = np.zeros((nl, self.N), dtype=np.uint64)
        self.L.hyo_encode(self.h, _ptr(slots), len(slots), self.delta if scale is None else scale, nl, _ptr(out))
        return out

    def encode_coeffs(self, slots, scale=None):
        slots = np.ascontiguousarray(slots, dtype=np.float64)
        out = np.zeros(self.N, dtype=np.int64)
        self.L.hyo_encode_coeffs(self.h, _ptr(slots), len(slots), self.delta if scale is None else scale, _ptr(out))
        return out

    def decode(self, poly_coeff, scale=None):
        poly_coeff = np.ascontiguousarray(poly_coeff, dtype=np.uint64)
        nl = poly_coeff.shape[0]
        out = np.zeros(self.slots, dtype=np.float64)
        self.L.hyo_decode(self.h, _ptr(poly_coeff), nl, self.delta if scale is None else scale, _ptr(out))
        return out


def sample_uniform(seed, stream, q, n):
    out = np.zeros(n, dtype=np.uint64)
    lib().hyo_sample_uniform(_ptr(seed_bytes(seed)), stream, int(q), _ptr(out), n)
    return out


def sample_ternary(seed, stream, n):
    out = np.zeros(n, dtype=np.int8)
    lib().hyo_sample_ternary(_ptr(seed_bytes(seed)), stream, _ptr(out), n)
    return out


def sample_gauss(seed, stream, n):
    out = np.zeros(n, dtype=np.int32)
    lib().hyo_sample_gauss(_ptr(seed_bytes(seed)), stream, _ptr(out), n)
    return out


def chacha_block(seed, stream, block):
    out = np.zeros(16, dtype=np.uint32)
    lib().hyo_chacha_block(_ptr(seed_bytes(seed)), stream, block, _ptr(out))
    return out


class Ct:
    """Owning wrapper of an oracle ciphertext."""

    def __init__(self, P, h, own=True):
        self.P, self.h, self.own = P, h, own

    def __del__(self):
        if getattr(self, "own", False) and self.h:
            self.P.L.hyo_ct_free(self.h)
            self.h = None

    @property
    def nl(self):
        return self.P.L.hyo_ct_nl(self.h)

    @property
    def npoly(self):
        return self.P.L.hyo_ct_npoly(self.h)

    @property
    def scale(self):
        return self.P.L.hyo_ct_scale(self.h)

    def data(self):
        """numpy view [npoly][nl][N] (no copy; valid while the ciphertext lives)."""
        n = self.npoly * self.nl * self.P.N
        buf = (C.c_uint64 * n).from_address(self.P.L.hyo_ct_data(self.h))
        return np.frombuffer(buf, dtype=np.uint64).reshape(self.npoly, self.nl, self.P.N)

    def clone(self):
        return Ct(self.P, self.P.L.hyo_ct_clone(self.P.h, self.h))


class CtArray:
    def __init__(self, P, h, n):
        self.P, self.h, self.n = P, h, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        assert 0 <= i < self.n
        return Ct(self.P, self.P.L.hyo_ct_at(self.h, i), own=False)

    def __del__(self):
        if self.h:
            self.P.L.hyo_ct_array_free(self.h, self.n)
            self.h = None


def default_rotations(P):
    """Rotation set of src/main.cpp:195-206 restricted to what approach 5 uses: 1..dim-1 and the positive powers
    of two below the slot count (EvalSum)."""
    r = list(range(1, P.dim))
    i = 1
    while i < P.slots:
        if i not in r:
            r.append(i)
        i *= 2
    return r


class Keys:
    def __init__(self, P, seed, rotations=None):
        self.P = P
        rot = default_rotations(P) if rotations is None else list(rotations)
        self.rotations = rot
        arr = np.array(rot if rot else [0], dtype=np.int32)
        self.seed = seed_bytes(seed)
        self.h = P.L.hyo_keygen(P.h, _ptr(self.seed), _ptr(arr), len(rot))

    def __del__(self):
        if getattr(self, "h", None):
            self.P.L.hyo_keys_free(self.h)
            self.h = None

    def _field(self, off_words, shape):
        raise NotImplementedError

    def rot_key(self, r):
        """numpy view [dnum][2][nT][N] of rotation key r."""
        P = self.P
        ptr = P.L.hyo_keys_rot(self.h, r)
        assert ptr, "no rotation key %d" % r
        n = P.dnum * 2 * P.nT * P.N
        buf = (C.c_uint64 * n).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint64).reshape(P.dnum, 2, P.nT, P.N)

    def _struct(self):
        class K(C.Structure):
            _fields_ = [("s_coeff", C.c_void_p), ("s_ntt", C.c_void_p), ("pk", C.c_void_p), ("relin", C.c_void_p),
                        ("n_rot", C.c_int), ("rot_idx", C.c_void_p), ("rot", C.c_void_p)]
        return K.from_address(self.h)

    def s_coeff(self):
        buf = (C.c_int8 * self.P.N).from_address(self._struct().s_coeff)
        return np.frombuffer(buf, dtype=np.int8)

    def s_ntt(self):
        P = self.P
        buf = (C.c_uint64 * (P.nT * P.N)).from_address(self._struct().s_ntt)
        return np.frombuffer(buf, dtype=np.uint64).reshape(P.nT, P.N)

    def pk(self):
        P = self.P
        buf = (C.c_uint64 * (2 * P.nQ * P.N)).from_address(self._struct().pk)
        return np.frombuffer(buf, dtype=np.uint64).reshape(2, P.nQ, P.N)

    def relin(self):
        P = self.P
        n = P.dnum * 2 * P.nT * P.N
        buf = (C.c_uint64 * n).from_address(self._struct().relin)
        return np.frombuffer(buf, dtype=np.uint64).reshape(P.dnum, 2, P.nT, P.N)


class Oracle:
    """Convenience front: the reference's role methods on the oracle."""

    def __init__(self, P, keys):
        self.P, self.K, self.L = P, keys, P.L

    def encrypt(self, slots, seed, nonce):
        slots = np.ascontiguousarray(slots, dtype=np.float64)
        return Ct(self.P, self.L.hyo_encrypt(self.P.h, self.K.h, _ptr(slots), len(slots), _ptr(seed_bytes(seed)), nonce))

    def decrypt(self, ct):
        out = np.zeros(self.P.slots, dtype=np.float64)
        self.L.hyo_decrypt(self.P.h, self.K.h, ct.h, _ptr(out))
        return out

    def rotate(self, ct, r):
        h = self.L.hyo_rotate(self.P.h, self.K.h, ct.h, r)
        assert h
        return Ct(self.P, h)

    def mult(self, a, b):
        return Ct(self.P, self.L.hyo_mult(self.P.h, self.K.h, a.h, b.h))

    def mult_norelin(self, a, b):
        return Ct(self.P, self.L.hyo_mult_norelin(self.P.h, a.h, b.h))

    def relin(self, a):
        self.L.hyo_relin_inplace(self.P.h, self.K.h, a.h)

    def rescale(self, a):
        self.L.hyo_rescale_inplace(self.P.h, a.h)

    def add(self, a, b):
        self.L.hyo_add_inplace(self.P.h, a.h, b.h)

    def add_const(self, a, c):
        self.L.hyo_add_const(self.P.h, a.h, c)

    def chebyshev_compare(self, ct, delta=0.44, depth=10):
        return Ct(self.P, self.L.hyo_chebyshev_compare(self.P.h, self.K.h, ct.h, delta, depth))

    # ---- roles
    def babies_for(self, n, matvec=None):
        """matvec: None = the product's auto rule, "hoisted" (the reference's form: every rotation hoisted), "bsgs" (the classic
        square-root split) or an explicit baby count (a power of two dividing vector_dim)"""
        if matvec is None:
            return auto_babies(self.P.dim, -(-n // self.P.slots))
        if matvec == "hoisted":
            return self.P.dim
        if matvec == "bsgs":
            return int(self.L.hyo_bsgs_babies(self.P.h))
        return int(matvec)

    def enroll(self, db, seed, matvec=None):
        """DiagonalEnroller::serializeDB — normalises `db` in place (like the reference).  The returned array remembers its baby
        count (`.babies`; vector_dim = the reference's hoisted form, less = pre-rotated diagonals for the baby-step / giant-step
        mat-vec); the sender methods follow it."""
        assert db.dtype == np.float64 and db.flags.c_contiguous and db.shape[1] == self.P.dim
        B = self.babies_for(db.shape[0], matvec)
        n_out = C.c_size_t(0)
        if B >= self.P.dim:
            h = self.L.hyo_enroll(self.P.h, self.K.h, _ptr(db), db.shape[0], _ptr(seed_bytes(seed)), C.byref(n_out))
        else:
            h = self.L.hyo_enroll_bsgs(self.P.h, self.K.h, _ptr(db), db.shape[0], _ptr(seed_bytes(seed)), C.byref(n_out), B)
        arr = CtArray(self.P, h, n_out.value)
        arr.babies = B
        arr.bsgs = B < self.P.dim
        return arr

    def encrypt_query(self, query, seed, nonce=1):
        query = np.ascontiguousarray(query, dtype=np.float64)
        return Ct(self.P, self.L.hyo_encrypt_query(self.P.h, self.K.h, _ptr(query), _ptr(seed_bytes(seed)), nonce))

    def rotate_query(self, q):
        return CtArray(self.P, self.L.hyo_rotate_query(self.P.h, self.K.h, q.h), self.P.dim)

    def compute_similarity(self, q, db, n):
        n_out = C.c_size_t(0)
        if getattr(db, "bsgs", False):
            h = self.L.hyo_compute_similarity_bsgs(self.P.h, self.K.h, q.h, db.h, n, C.byref(n_out), db.babies)
        else:
            h = self.L.hyo_compute_similarity(self.P.h, self.K.h, q.h, db.h, n, C.byref(n_out))
        return CtArray(self.P, h, n_out.value)

    def index_scenario(self, q, db, n):
        n_out = C.c_size_t(0)
        if getattr(db, "bsgs", False):
            h = self.L.hyo_index_scenario_bsgs(self.P.h, self.K.h, q.h, db.h, n, C.byref(n_out), db.babies)
        else:
            h = self.L.hyo_index_scenario(self.P.h, self.K.h, q.h, db.h, n, C.byref(n_out))
        return CtArray(self.P, h, n_out.value)

    def membership_scenario(self, q, db, n):
        if getattr(db, "bsgs", False):
            return Ct(self.P, self.L.hyo_membership_scenario_bsgs(self.P.h, self.K.h, q.h, db.h, n, db.babies))
        return Ct(self.P, self.L.hyo_membership_scenario(self.P.h, self.K.h, q.h, db.h, n))

    def write_db_files(self, db, directory):
        """one index<t>.bin per ciphertext (enroller_diag.cpp:158-166)"""
        assert self.L.hyo_db_write_files(self.P.h, db.h, len(db), str(directory).encode()) == 0

    def index_scenario_files(self, q, directory, n):
        """indexScenario that re-reads every database ciphertext from disk inside loop B (sender_diag.cpp:85-94)"""
        n_out = C.c_size_t(0)
        h = self.L.hyo_index_scenario_files(self.P.h, self.K.h, q.h, str(directory).encode(), n, C.byref(n_out))
        return CtArray(self.P, h, n_out.value)

    # ---- HERS (approach 4)
    def hers_enroll(self, db, seed):
        assert db.dtype == np.float64 and db.flags.c_contiguous and db.shape[1] == self.P.dim
        n_out = C.c_size_t(0)
        h = self.L.hyo_hers_enroll(self.P.h, self.K.h, _ptr(db), db.shape[0], _ptr(seed_bytes(seed)), C.byref(n_out))
        return CtArray(self.P, h, n_out.value)

    def hers_encrypt_query(self, query, seed, nonce0=1000):
        query = np.ascontiguousarray(query, dtype=np.float64)
        h = self.L.hyo_hers_encrypt_query(self.P.h, self.K.h, _ptr(query), _ptr(seed_bytes(seed)), nonce0)
        return CtArray(self.P, h, self.P.dim)

    def hers_compute_similarity(self, q, db, n):
        n_out = C.c_size_t(0)
        h = self.L.hyo_hers_compute_similarity(self.P.h, self.K.h, q.h, db.h, n, C.byref(n_out))
        return CtArray(self.P, h, n_out.value)

    def hers_index_scenario(self, q, db, n):
        n_out = C.c_size_t(0)
        h = self.L.hyo_hers_index_scenario(self.P.h, self.K.h, q.h, db.h, n, C.byref(n_out))
        return CtArray(self.P, h, n_out.value)

    def hers_membership_scenario(self, q, db, n):
        return Ct(self.P, self.L.hyo_hers_membership_scenario(self.P.h, self.K.h, q.h, db.h, n))

    def decrypt_membership(self, ct):
        return bool(self.L.hyo_decrypt_membership(self.P.h, self.K.h, ct.h))

    def decrypt_index(self, cts):
        cap = len(cts) * self.P.slots
        out = np.zeros(cap, dtype=np.uint64)
        n = self.L.hyo_decrypt_index(self.P.h, self.K.h, cts.h, len(cts), _ptr(out), cap)
        return [int(v) for v in out[:n]]


def read_dat(path):
    """Dataset format of the reference's test/*.dat (src/main.cpp:56-57, :216-230): n, the query, n rows."""
    with open(path) as f:
        tok = f.read().split()
    n = int(tok[0])
    vals = np.array(tok[1:], dtype=np.float64)
    dim = (len(vals)) // (n + 1)
    return n, vals[:dim].copy(), vals[dim:].reshape(n, dim).copy()


def alt_prime_chain(log_n, n_q=12, n_p=4, scale_bits=45, skip=25):
    """A prime chain DIFFERENT from the derived one, with the shape OpenFHE produces for the reference's context (one ~60-bit
    first prime, n_q-1 primes around 2^scale_bits, n_p ~60-bit special primes) — stands in for an externally generated
    chain in the custom-context tests.  Returns (moduli, roots): roots are psi^3 of the smallest primitive 2N-th root, i.e.
    also not the default choice."""
    from sympy import isprime
    M = 2 << log_n

    def walk(start, step, count, skip_first):
        out, c = [], start - (start % M) + 1
        while len(out) < count + skip_first:
            c += step * M
            if isprime(c):
                out.append(c)
        return out[skip_first:]
    scal = walk(1 << scale_bits, +1, (n_q - 1 + 1) // 2, skip) + walk(1 << scale_bits, -1, (n_q - 1) // 2, skip)
    q = walk((1 << 59) + (1 << 58), -1, 1, 3) + scal[:n_q - 1]
    p = walk(1 << 59, +1, n_p, 5)
    moduli = q + p
    roots = []
    for m in moduli:
        x = 2
        while True:
            r = pow(x, (m - 1) // M, m)
            if pow(r, M // 2, m) == m - 1:
                break
            x += 1
        roots.append(pow(r, 3, m))
    return np.array(moduli, dtype=np.uint64), np.array(roots, dtype=np.uint64)
