//! arkworks values <-> the byte layout of `include/dg16.h` ("Conventions").
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ff::{BigInt, Field, Fp, MontBackend, MontConfig, PrimeField};

/// Montgomery limbs of a prime-field element as they lie in memory: `Fp<MontBackend<_, N>>.0` is `BigInt<N>([u64; N])`
/// (pinned by ark-circom/src/zkey.rs:417-427, which reads the same limbs from a `.zkey`).
#[inline]
pub fn fp_limbs<T: MontConfig<N>, const N: usize>(x: &Fp<MontBackend<T, N>, N>) -> &[u64; N] {
    &x.0 .0
}
#[inline]
pub fn fp_from_limbs<T: MontConfig<N>, const N: usize>(l: [u64; N]) -> Fp<MontBackend<T, N>, N> {
    Fp::new_unchecked(BigInt::new(l))
}

/// Base-field elements (prime field, or the quadratic extension as `c0 || c1`) to little-endian Montgomery bytes.
pub trait FieldBytes: Field {
    const BYTES: usize;
    fn write_mont(&self, out: &mut [u8]);
    fn read_mont(inp: &[u8]) -> Self;
}
impl<T: MontConfig<N>, const N: usize> FieldBytes for Fp<MontBackend<T, N>, N> {
    const BYTES: usize = 8 * N;
    fn write_mont(&self, out: &mut [u8]) {
        for (i, l) in fp_limbs(self).iter().enumerate() {
            out[8 * i..8 * i + 8].copy_from_slice(&l.to_le_bytes());
        }
    }
    fn read_mont(inp: &[u8]) -> Self {
        let mut l = [0u64; N];
        for i in 0..N {
            l[i] = u64::from_le_bytes(inp[8 * i..8 * i + 8].try_into().unwrap());
        }
        fp_from_limbs(l)
    }
}
impl<P: ark_ff::Fp2Config> FieldBytes for ark_ff::Fp2<P>
where
    P::Fp: FieldBytes,
{
    const BYTES: usize = 2 * <P::Fp as FieldBytes>::BYTES;
    fn write_mont(&self, out: &mut [u8]) {
        let h = Self::BYTES / 2;
        self.c0.write_mont(&mut out[..h]);
        self.c1.write_mont(&mut out[h..]);
    }
    fn read_mont(inp: &[u8]) -> Self {
        let h = Self::BYTES / 2;
        ark_ff::Fp2::<P>::new(<P::Fp>::read_mont(&inp[..h]), <P::Fp>::read_mont(&inp[h..]))
    }
}

/// `x || y`, identity = all-zero bytes (the convention of ark-circom/src/zkey.rs:353-361): 64 / 96 bytes per G1 point,
/// 128 / 192 per G2 point.
pub fn pack_affine<P: SWCurveConfig>(pts: &[Affine<P>]) -> Vec<u8>
where
    P::BaseField: FieldBytes,
{
    let fe = <P::BaseField as FieldBytes>::BYTES;
    let mut out = vec![0u8; pts.len() * 2 * fe];
    for (i, p) in pts.iter().enumerate() {
        if p.infinity {
            continue;
        }
        p.x.write_mont(&mut out[i * 2 * fe..][..fe]);
        p.y.write_mont(&mut out[i * 2 * fe + fe..][..fe]);
    }
    out
}

/// Jacobian `(x, y, z)` Montgomery limbs (what every group-valued entry point writes) -> `Projective`.
/// z = 0 encodes the identity, like ark-ec.
pub fn unpack_projective<P: SWCurveConfig>(buf: &[u8]) -> Projective<P>
where
    P::BaseField: FieldBytes,
{
    let fe = <P::BaseField as FieldBytes>::BYTES;
    Projective::new_unchecked(
        <P::BaseField>::read_mont(&buf[..fe]),
        <P::BaseField>::read_mont(&buf[fe..2 * fe]),
        <P::BaseField>::read_mont(&buf[2 * fe..3 * fe]),
    )
}

/// `&[F]` of a 256-bit scalar field as the bytes the library reads with `DG16_F_SCALARS_MONT` -- no copy.
pub fn scalars_as_bytes<F: PrimeField>(s: &[F]) -> &[u8] {
    assert_eq!(core::mem::size_of::<F>(), 32, "Fr of BN254 / BLS12-381 / BLS12-377 is 4 x u64");
    unsafe { core::slice::from_raw_parts(s.as_ptr().cast::<u8>(), s.len() * 32) }
}
pub fn scalars_as_bytes_mut<F: PrimeField>(s: &mut [F]) -> &mut [u8] {
    assert_eq!(core::mem::size_of::<F>(), 32);
    unsafe { core::slice::from_raw_parts_mut(s.as_mut_ptr().cast::<u8>(), s.len() * 32) }
}
