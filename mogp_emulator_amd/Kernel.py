"""
Kernel name objects.  The reference GPU wrapper keeps a *Python* kernel instance in
``gp.kernel`` purely for interface compatibility (GaussianProcessGPU.py:263-277) while all
covariance arithmetic runs in the native backend; the same holds here, so these classes carry
the identity of the kernel and the parameter count, not a host implementation of it.
"""
import numpy as np


class KernelBase(object):
    native_name = None

    def get_n_params(self, inputs):
        """One correlation length per input dimension (Kernel.py:419-442)."""
        inputs = np.asarray(inputs)
        return 1 if inputs.ndim == 1 else int(inputs.shape[1])

    def kernel_f(self, x1, x2, params):
        """sigma^2-free kernel matrix evaluated ON THE DEVICE (square case x1 is x2)."""
        from . import libgpgpu
        params = np.asarray(params, dtype=np.float64)
        k = getattr(libgpgpu, self.native_name + "Kernel")()
        n = np.atleast_2d(x1).shape[0]
        return k.kernel_f(x1, x2, np.append(params, 0.)).reshape(n, -1)

    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self).__name__)


class SquaredExponential(KernelBase):
    native_name = "SquaredExponential"

    def __str__(self):
        return "Squared Exponential Kernel"


class Matern52(KernelBase):
    native_name = "Matern52"

    def __str__(self):
        return "Matern 5/2 Kernel"


class ProductMat52(KernelBase):
    """product over the inputs of one-dimensional Matern-5/2 kernels (Kernel.py:581-763, 986-997)"""
    native_name = "ProductMat52"

    def __str__(self):
        return "Product Matern 5/2 Kernel"


class UniformSqExp(KernelBase):
    """squared exponential with ONE correlation length shared by all inputs (Kernel.py:224-417, 956-964)"""
    native_name = "UniformSqExp"

    def get_n_params(self, inputs):
        return 1

    def __str__(self):
        return "Uniform Squared Exponential Kernel"


class UniformMat52(UniformSqExp):
    native_name = "UniformMat52"

    def __str__(self):
        return "Uniform Matern 5/2 Kernel"
