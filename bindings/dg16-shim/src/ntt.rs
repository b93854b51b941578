//! `Radix2EvaluationDomain::{fft,ifft}_in_place` -> `dg16_ntt`; the transform part of
//! `CircomReduction::witness_map_from_matrices` (ark-circom/src/circom/qap.rs:64-91) -> `dg16_h_poly`.
use crate::pack::{scalars_as_bytes, scalars_as_bytes_mut};
use crate::{check, sys, Dg16Error, CTX};
use ark_ff::PrimeField;
use std::os::raw::{c_int, c_uint};

fn log2_exact(n: usize) -> Result<c_uint, Dg16Error> {
    if n.is_power_of_two() {
        Ok(n.trailing_zeros())
    } else {
        Err(Dg16Error::Status(sys::DG16_ERR_BAD_ARG, format!("domain size {} is not a power of two", n)))
    }
}

/// `domain.fft_in_place(v)` (inverse = false) / `domain.ifft_in_place(v)` (inverse = true, scaled by 1/n) for a
/// `Radix2EvaluationDomain` of `v.len()` points; `coset_offset = Some(g)` is `domain.get_coset(g)`: forward multiplies
/// coefficient i by g^i first, inverse multiplies output i by g^-i.  Natural order in and out.
pub fn ntt_in_place<F: PrimeField>(
    curve: c_int,
    v: &mut [F],
    inverse: bool,
    coset_offset: Option<F>,
    channel: c_int,
) -> Result<(), Dg16Error> {
    let log_n = log2_exact(v.len())?;
    let off = coset_offset.map(|g| [g]);
    let off_ptr = off.as_ref().map_or(core::ptr::null(), |g| scalars_as_bytes(&g[..]).as_ptr());
    check(unsafe {
        sys::dg16_ntt(
            CTX.0,
            curve,
            scalars_as_bytes_mut(v).as_mut_ptr().cast(),
            log_n,
            inverse as c_int,
            off_ptr.cast(),
            0,
            channel,
        )
    })
}

/// Lines 64-91 of `witness_map_from_matrices`: from the evaluation vectors a, b, c (each `domain_size` long, c = a o b
/// on the constraint rows) to `ab - c` on the coset: three inverse transforms, the w_{2m}^i shift, three forward
/// transforms and the pointwise product in ONE call, a / b / c never leave HBM in between.
pub fn h_poly<F: PrimeField>(curve: c_int, a: &[F], b: &[F], c: &[F], channel: c_int) -> Result<Vec<F>, Dg16Error> {
    if a.len() != b.len() || a.len() != c.len() {
        return Err(Dg16Error::LengthMismatch(a.len().min(b.len()).min(c.len())));
    }
    let log_m = log2_exact(a.len())?;
    let mut h = vec![F::zero(); a.len()];
    check(unsafe {
        sys::dg16_h_poly(
            CTX.0,
            curve,
            scalars_as_bytes(a).as_ptr().cast(),
            scalars_as_bytes(b).as_ptr().cast(),
            scalars_as_bytes(c).as_ptr().cast(),
            log_m,
            scalars_as_bytes_mut(&mut h).as_mut_ptr().cast(),
            0,
            channel,
        )
    })?;
    Ok(h)
}
