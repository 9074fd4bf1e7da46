"""sunscreen_amd -- MI355X-native BFV ciphertext-arithmetic backend (the path behind seal_fhe::Evaluator).

Layers:
  csrc/           hand-written HIP kernels + host runtime + the C ABI (libhipbfv.so, include/hipbfv.h)
  _lib.py         ctypes loader for the C ABI (fails loudly when the library is missing)
  seal.py         host-side mirror of the `seal_fhe` crate surface for this path (Context, BFVEvaluator,
                  Ciphertext, Plaintext, RelinearizationKeys, GaloisKeys, ...)
  batch.py        the GPU batch executor: the same operations over device-resident batches of ciphertexts
"""
from .seal import (  # noqa: F401
    BFVEncoder,
    BFVEvaluator,
    BfvEncryptionParametersBuilder,
    Ciphertext,
    CoefficientModulus,
    Context,
    Decryptor,
    Encryptor,
    GaloisKeys,
    HipBfvError,
    KeyGenerator,
    Modulus,
    PlainModulus,
    Plaintext,
    PolynomialArray,
    PublicKey,
    RelinearizationKeys,
    SecretKey,
    SecurityLevel,
)
