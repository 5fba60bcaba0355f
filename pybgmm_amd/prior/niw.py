"""Normal-inverse-Wishart prior: the parameter bag of reference pybgmm/prior/niw.py:8-23."""


class NIW(object):
    """A normal-inverse-Wishart distribution ``NIW(m_0, k_0, v_0, S_0)``.

    m_0: prior mean of the component mean; k_0: belief in m_0; v_0: degrees of
    freedom (belief in S_0; must be >= D, and integer valued for the samplers);
    S_0: proportional to the prior mean of the covariance.
    """

    def __init__(self, m_0, k_0, v_0, S_0):
        self.m_0 = m_0
        self.k_0 = k_0
        D = len(m_0)
        assert v_0 >= D, "v_0 must be larger or equal to dimension of data"
        self.v_0 = v_0
        self.S_0 = S_0
