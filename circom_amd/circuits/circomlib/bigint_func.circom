/*
    Long arithmetic on k limbs of n bits as circom FUNCTIONS, in the shape of 0xPARC circom-ecdsa's bigint_func.circom
    (circom-ecdsa is not part of the reference tree; the algorithms are the published ones: Knuth D with the
    normalisation 2^n \ (1 + b[k-1]), schoolbook products with deferred carries).  A template calls them with SIGNAL
    values: the comparisons below are run-time branches, the loops over limbs are compile-time loops.
    Arrays carry their exact sizes (circom-ecdsa over-allocates 100 entries everywhere; a register per entry is what a
    run-time function costs here).  The circom text of circom_amd/circuits/bigint.py.
*/
pragma circom 2.0.0;

// 1 when a > b on k limbs
function long_gt(n, k, a, b) {
    for (var i = k - 1; i >= 0; i--) {
        if (a[i] > b[i]) {
            return 1;
        }
        if (a[i] < b[i]) {
            return 0;
        }
    }
    return 0;
}

// a - b on k limbs, a >= b
function long_sub(n, k, a, b) {
    var diff[k];
    var borrow[k];
    for (var i = 0; i < k; i++) {
        if (i == 0) {
            if (a[i] >= b[i]) {
                diff[i] = a[i] - b[i];
                borrow[i] = 0;
            } else {
                diff[i] = a[i] - b[i] + (1 << n);
                borrow[i] = 1;
            }
        } else {
            if (a[i] >= b[i] + borrow[i - 1]) {
                diff[i] = a[i] - b[i] - borrow[i - 1];
                borrow[i] = 0;
            } else {
                diff[i] = (1 << n) + a[i] - b[i] - borrow[i - 1];
                borrow[i] = 1;
            }
        }
    }
    return diff;
}

// a * b for a single-limb a and a k-limb b: k + 1 limbs
function long_scalar_mult(n, k, a, b) {
    var out[k + 1];
    for (var i = 0; i < k; i++) {
        var temp = out[i] + (a * b[i]);
        out[i] = temp % (1 << n);
        out[i + 1] = out[i + 1] + temp \ (1 << n);
    }
    return out;
}

// quotient digit of a (k + 1 limbs) by b (k limbs, top limb >= 2^(n-1))
function short_div_norm(n, k, a, b) {
    var qhat = (a[k] * (1 << n) + a[k - 1]) \ b[k - 1];
    if (qhat > (1 << n) - 1) {
        qhat = (1 << n) - 1;
    }
    var bext[k + 1];
    for (var i = 0; i < k; i++) {
        bext[i] = b[i];
    }
    var mult[k + 1] = long_scalar_mult(n, k, qhat, b);
    if (long_gt(n, k + 1, mult, a) == 1) {
        mult = long_sub(n, k + 1, mult, bext);
        if (long_gt(n, k + 1, mult, a) == 1) {
            return qhat - 2;
        } else {
            return qhat - 1;
        }
    } else {
        return qhat;
    }
}

// quotient digit of a (k + 1 limbs) by b (k limbs), a < 2^n * b
function short_div(n, k, a, b) {
    var scale = (1 << n) \ (1 + b[k - 1]);
    var norm_a[k + 2] = long_scalar_mult(n, k + 1, scale, a);
    var norm_b[k + 1] = long_scalar_mult(n, k, scale, b);
    return short_div_norm(n, k, norm_a, norm_b);
}

// a (k + m limbs) = out[0] (m + 1 limbs) * b (k limbs) + out[1] (k limbs), b[k - 1] != 0
function long_div(n, k, m, a, b) {
    var w = m + 1 > k ? m + 1 : k;
    var out[2][w];
    var remainder[m + k];
    for (var i = 0; i < m + k; i++) {
        remainder[i] = a[i];
    }
    var dividend[k + 1];
    for (var i = m; i >= 0; i--) {
        if (i == m) {
            dividend[k] = 0;
            for (var j = k - 1; j >= 0; j--) {
                dividend[j] = remainder[j + m];
            }
        } else {
            for (var j = k; j >= 0; j--) {
                dividend[j] = remainder[j + i];
            }
        }
        out[0][i] = short_div(n, k, dividend, b);
        var mult_shift[k + 1] = long_scalar_mult(n, k, out[0][i], b);
        var subtrahend[m + k];
        for (var j = 0; j <= k; j++) {
            if (i + j < m + k) {
                subtrahend[i + j] = mult_shift[j];
            }
        }
        remainder = long_sub(n, m + k, remainder, subtrahend);
    }
    for (var i = 0; i < k; i++) {
        out[1][i] = remainder[i];
    }
    return out;
}

// a * b on k limbs: 2k limbs (column sums first - k products of 2n bits stay far below the field's size - then one carry pass)
function prod(n, k, a, b) {
    var prod_val[2 * k - 1];
    for (var i = 0; i < 2 * k - 1; i++) {
        if (i < k) {
            for (var a_idx = 0; a_idx <= i; a_idx++) {
                prod_val[i] = prod_val[i] + a[a_idx] * b[i - a_idx];
            }
        } else {
            for (var a_idx = i - k + 1; a_idx < k; a_idx++) {
                prod_val[i] = prod_val[i] + a[a_idx] * b[i - a_idx];
            }
        }
    }
    var out[2 * k];
    var carry = 0;
    for (var i = 0; i < 2 * k - 1; i++) {
        var t = prod_val[i] + carry;
        out[i] = t % (1 << n);
        carry = t \ (1 << n);
    }
    out[2 * k - 1] = carry;
    return out;
}

// a * b mod p on k limbs
function prod_mod(n, k, a, b, p) {
    var t[2 * k] = prod(n, k, a, b);
    var qr[2][k + 1] = long_div(n, k, k, t, p);
    var out[k];
    for (var i = 0; i < k; i++) {
        out[i] = qr[1][i];
    }
    return out;
}

// a^e mod p by square-and-multiply from the top bit of e down: n * k trips whose body is two long products and divisions -
// the compiler leaves the loop to run time after the first trips (its counter then indexes eBits at run time)
function mod_exp(n, k, a, p, e) {
    var eBits[n * k];
    for (var i = 0; i < k; i++) {
        for (var j = 0; j < n; j++) {
            eBits[i * n + j] = (e[i] >> j) & 1;
        }
    }
    var out[k];
    out[0] = 1;
    for (var i = k * n - 1; i >= 0; i--) {
        out = prod_mod(n, k, out, out, p);
        if (eBits[i] == 1) {
            out = prod_mod(n, k, out, a, p);
        }
    }
    return out;
}

// a^(p - 2) mod p for a prime p (0 for a = 0)
function mod_inv(n, k, a, p) {
    var isZero = 1;
    for (var i = 0; i < k; i++) {
        if (a[i] != 0) {
            isZero = 0;
        }
    }
    var zeros[k];
    if (isZero == 1) {
        return zeros;
    }
    var two[k];
    two[0] = 2;
    var pm2[k] = long_sub(n, k, p, two);
    return mod_exp(n, k, a, p, pm2);
}
